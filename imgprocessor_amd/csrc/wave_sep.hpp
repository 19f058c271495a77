// wave_sep.hpp - separable K+K correlation (scipy.ndimage.gaussian_filter order: axis 0
// then axis 1, intermediate rounded to float32) on the wave-marching skeleton of
// wave_stencil.hpp, float32, K = 3, 5, 7, 9 taps on both axes.
//
// A wave64 owns a 256-px-wide strip.  Each arriving input row is scattered into the K
// running y-sums it belongs to (K x 4 registers per lane, shifted inside the fma
// chain); the y-row that completes is already the float32 intermediate scipy
// stores between its two passes; its x pass needs K/2 neighbours per side,
// taken from the adjacent lanes with DPP wave shifts, and the result is stored
// as one float4.  2K fma per pixel instead of K*K, one pass over HBM, no barriers.
//
// Row sources: plain image rows (LoadRowSrc: rows stay in registers) or rows of a remapped
// image sampled on the fly (SampleRowSrc: blended samples pass through the wave-private LDS
// row of wave_stencil.hpp, which also undoes the lane-interleaved sampling order) - the
// remap -> Gaussian chain of PerspectiveCorrection.correct / LensDistortion.correct followed
// by scipy.ndimage.gaussian_filter in one kernel.
// Reference call sites: filters/standardDeviation.py:23, filters/fastFilter.py:42,
// camera/flatField/flatField.py:47.
#pragma once

#include "common.hpp"
#include "wave_stencil.hpp"

namespace ipa {

// strip geometry of the separable kernels: as wave_geom (the 240-px step of the short kernels)
#ifndef IPA_SEP_MIN_HL
#define IPA_SEP_MIN_HL 2
#endif
template <int K> struct sep_geom {
  static constexpr int HL = (K / 2 + 3) / 4 > IPA_SEP_MIN_HL ? (K / 2 + 3) / 4 : IPA_SEP_MIN_HL;
  static constexpr int OW = 256 - 8 * HL;
};
// K = 1: no filter at all - the remap alone on the marching strips (round 6: "strip remap").  No halo lanes, no
// halo rows: 256-px strips whose rows are whole 1024-byte line runs; both passes are the identity (no multiply).
template <> struct sep_geom<1> {
  static constexpr int HL = 0;
  static constexpr int OW = 256;
};

template <int K> struct SepTaps {
  float ky[K], kx[K];
};

// chunk rows.  Sampling sources with analytic (f64) coordinates take one row at a time here:
// two rows of coordinates next to K x 4 running sums cost 190-200 VGPRs (occupancy 2) and
// measured 13 % slower on the 4K perspective + 9-tap chain.
template <typename Src, int K> struct sep_depth {
  static constexpr int value = Src::kMap ? Src::template depth<K>::value : 1;
};
template <int K> struct sep_depth<LoadRowSrc, K> { static constexpr int value = 4; };

// The two passes in packed float32 (round 6): a lane's four pixels are two register pairs, one v_pk_fma_f32 per tap
// advances a pair - K + K packed fmas per row and lane instead of 4 K + 4 K scalar ones (the taps are halves of SGPR
// pairs broadcast by op_sel, as in the dense loops).  Each half of a packed fma is the scalar fma of the same
// operands in the same order: same bits as the scalar form (IPA_SEP_PACKED = 0 builds that one).
#ifndef IPA_SEP_PACKED
#define IPA_SEP_PACKED 1
#endif
template <int K, int I> __device__ __forceinline__ v2f sep_tap_pair(const float (&k)[K]) {
  constexpr int n0 = I & ~1, n1 = n0 + 1 < K ? n0 + 1 : n0;
  return v2f{k[n0], k[n1]};
}
// y pass: the arriving row (c0, c1 = pixel pairs) feeds the K running intermediate rows; acc[K - 1] completes
template <int K>
__device__ __forceinline__ void sep_y_pass(const SepTaps<K>& w, v2f (&acc)[K][2], v2f c0, v2f c1) {
  if constexpr (K == 1) {   // the remap alone
    acc[0][0] = c0;
    acc[0][1] = c1;
    return;
  }
  static_for<0, K>([&](auto Ii) {
    constexpr int i = K - 1 - decltype(Ii)::value;
    const v2f t = sep_tap_pair<K, i>(w.ky);
    if constexpr (i == 0) {
      acc[0][0] = pk_mul_coef<(i & 1)>(t, c0);
      acc[0][1] = pk_mul_coef<(i & 1)>(t, c1);
    } else {
      acc[i][0] = pk_fma_coef<(i & 1)>(t, c0, acc[i - 1][0]);
      acc[i][1] = pk_fma_coef<(i & 1)>(t, c1, acc[i - 1][1]);
    }
  });
}
// x pass on a completed intermediate row: K / 2 neighbours per side from the adjacent lanes (DPP wave shifts)
template <int K>
__device__ __forceinline__ v4f sep_x_pass(const SepTaps<K>& w, const float (&mid)[4]) {
  constexpr int H = K / 2;
  if constexpr (K == 1) return v4f{mid[0], mid[1], mid[2], mid[3]};   // the remap alone
  static_assert(K == 1 || (H >= 1 && H <= 4), "3 .. 9 taps");
  // win[m] = intermediate pixel 4 L - H + m, m = 0 .. 3 + 2 H, held as the pairs (2 n, 2 n + 1) when H is even and
  // (2 n - 1, 2 n) when it is odd, so that the lane's own four pixels are its own two register pairs either way
  float win[4 + 2 * H];
#pragma unroll
  for (int k = 0; k < 4; k++) win[H + k] = mid[k];
#pragma unroll
  for (int m = 0; m < H; m++) {
    win[H - 1 - m] = from_lane_below(mid[3 - m]);
    win[H + 4 + m] = from_lane_above(mid[m]);
  }
  v2f out[2];
  static_for<0, K>([&](auto Jj) {
    constexpr int j = decltype(Jj)::value;
    const v2f t = sep_tap_pair<K, j>(w.kx);
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const v2f x = v2f{win[2 * h + j], win[2 * h + j + 1]};
      if constexpr (j == 0) out[h] = pk_mul_coef<(j & 1)>(t, x);
      else out[h] = pk_fma_coef<(j & 1)>(t, x, out[h]);
    }
  });
  return v4f{out[0].x, out[0].y, out[1].x, out[1].y};
}

template <bool FAST, typename Src, int K, int QM = -1>
__device__ __forceinline__ void wave_sep_strip(const WaveParams& p, const Src& src,
                                               const SepTaps<K>& w, float* xp, const Cols& c,
                                               int y0, int nrows, bool writer, float* dst,
                                               float xcval) {
  constexpr int H = K / 2;
  constexpr int D = sep_depth<Src, K>::value;
  constexpr bool kRegs = std::is_same<Src, LoadRowSrc>::value;  // rows stay in registers
  const int T = nrows + K - 1;
  float acc[K][4];  // acc[i] = y-sum of intermediate row (t - i)
  v2f acc2[K][2];   // ... as pixel pairs (IPA_SEP_PACKED; only one of the two forms is live)
#pragma unroll 1
  for (int tb = 0; tb < T; tb += D) {
    int vv[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
      if constexpr (FAST) vv[d] = y0 - H + tb + d;
      else vv[d] = resolve_idx(y0 - H + tb + d, p.dh, p.cby);
    }
    typename Src::template Chunk<D> ch;
    src.template load_chunk<FAST, D, QM>(c, vv, ch);
    if constexpr (!kRegs) {
      src.template stage_rows<FAST, D>(c, vv, ch, xp);
      __builtin_amdgcn_wave_barrier();  // wave-private LDS rows: in-order ds_write / ds_read
    }

    static_for<0, D>([&](auto Dd) {
      constexpr int d = decltype(Dd)::value;
      const int t = tb + d;
      float cur[4];
      if constexpr (kRegs) {
#pragma unroll
        for (int k = 0; k < 4; k++) cur[k] = ch.v[d][k];
      } else {
        const float4 q = *reinterpret_cast<const float4*>(xp + d * kRowStride + kRowPad +
                                                          4u * (threadIdx.x & 63u));
        cur[0] = q.x; cur[1] = q.y; cur[2] = q.z; cur[3] = q.w;
      }

      // y pass: the arriving row feeds K intermediate rows
      if constexpr (IPA_SEP_PACKED != 0) {
        sep_y_pass<K>(w, acc2, v2f{cur[0], cur[1]}, v2f{cur[2], cur[3]});
      } else {
        static_for<0, K>([&](auto Ii) {
          constexpr int i = K - 1 - decltype(Ii)::value;
#pragma unroll
          for (int ox = 0; ox < 4; ox++) {
            if constexpr (i == 0) acc[0][ox] = w.ky[0] * cur[ox];
            else acc[i][ox] = fmaf(w.ky[i], cur[ox], acc[i - 1][ox]);
          }
        });
      }

      const int o = t - (K - 1);
      if (o >= 0 && o < nrows) {  // wave-uniform: intermediate row o is complete
        float mid[4];
        if constexpr (IPA_SEP_PACKED != 0) {
          mid[0] = acc2[K - 1][0].x; mid[1] = acc2[K - 1][0].y;
          mid[2] = acc2[K - 1][1].x; mid[3] = acc2[K - 1][1].y;
        } else {
#pragma unroll
          for (int k = 0; k < 4; k++) mid[k] = acc[K - 1][k];
        }
        if constexpr (!FAST) {
          // constant x border: scipy pads the INTERMEDIATE with cval
#pragma unroll
          for (int k = 0; k < 4; k++)
            if (c.uu[k] < 0) mid[k] = xcval;
        }
        float out[4];
        if constexpr (IPA_SEP_PACKED != 0) {
          const v4f q = sep_x_pass<K>(w, mid);
          out[0] = q.x; out[1] = q.y; out[2] = q.z; out[3] = q.w;
        } else {
          float win[4 + 2 * H];
#pragma unroll
          for (int k = 0; k < 4; k++) win[H + k] = mid[k];
#pragma unroll
          for (int m = 0; m < H; m++) {
            win[H - 1 - m] = from_lane_below(mid[3 - m]);
            win[H + 4 + m] = from_lane_above(mid[m]);
          }
#pragma unroll
          for (int ox = 0; ox < 4; ox++) {
            float a = w.kx[0] * win[ox];
#pragma unroll
            for (int j = 1; j < K; j++) a = fmaf(w.kx[j], win[ox + j], a);
            out[ox] = a;
          }
        }
        if (writer) {
          float* row = dst + (long)(y0 + o) * p.dpitch + c.xo;
          const int n = p.dw - c.xo < 4 ? p.dw - c.xo : 4;
          if constexpr (FAST) {
            float* rows_ = dst + ((long)(y0 + o) * p.dpitch + c.xs);  // scalar base
            *reinterpret_cast<float4*>(rows_ + 4u * (threadIdx.x & 63u)) =
                float4{out[0], out[1], out[2], out[3]};
          } else if (p.vec_out && n == 4) {
            *reinterpret_cast<float4*>(row) = float4{out[0], out[1], out[2], out[3]};
          } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
              if (k < n) row[k] = out[k];
          }
        }
      }
    });
    if constexpr (!kRegs) __builtin_amdgcn_wave_barrier();
  }
}

// the separable filter as a policy of wave_run_strip_shared (wave_pipe.hpp): the sample row
// arrives in the wave's LDS row; y pass into the K running intermediate rows, x pass on the
// completed one through DPP wave shifts - the arithmetic and order of wave_sep_strip
template <int K> struct SepFilter {
  const SepTaps<K>& w;
  float xcval;
  float acc[K][4];
  v2f acc2[K][2];   // (IPA_SEP_PACKED: the running rows as pixel pairs; only one of the two forms is live)
  __device__ __forceinline__ SepFilter(const SepTaps<K>& taps, float xc) : w(taps), xcval(xc) {}
  template <bool EDGE> __device__ __forceinline__ v4f row(const float* xp, unsigned lane, const Cols& c) {
    constexpr int H = K / 2;
    const float4 q = *reinterpret_cast<const float4*>(xp + kRowPad + 4u * lane);
    const float cur[4] = {q.x, q.y, q.z, q.w};
    if constexpr (IPA_SEP_PACKED != 0) {
      sep_y_pass<K>(w, acc2, v2f{cur[0], cur[1]}, v2f{cur[2], cur[3]});
      float mid[4] = {acc2[K - 1][0].x, acc2[K - 1][0].y, acc2[K - 1][1].x, acc2[K - 1][1].y};
      if constexpr (EDGE) {
        // constant x border: scipy pads the INTERMEDIATE with cval
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (c.uu[k] < 0) mid[k] = xcval;
      }
      return sep_x_pass<K>(w, mid);
    }
    static_for<0, K>([&](auto Ii) {
      constexpr int i = K - 1 - decltype(Ii)::value;
#pragma unroll
      for (int ox = 0; ox < 4; ox++) {
        if constexpr (i == 0) acc[0][ox] = w.ky[0] * cur[ox];
        else acc[i][ox] = fmaf(w.ky[i], cur[ox], acc[i - 1][ox]);
      }
    });
    float mid[4];
#pragma unroll
    for (int k = 0; k < 4; k++) mid[k] = acc[K - 1][k];
    if constexpr (EDGE) {
      // constant x border: scipy pads the INTERMEDIATE with cval
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (c.uu[k] < 0) mid[k] = xcval;
    }
    float win[4 + 2 * H];
#pragma unroll
    for (int k = 0; k < 4; k++) win[H + k] = mid[k];
#pragma unroll
    for (int m = 0; m < H; m++) {
      win[H - 1 - m] = from_lane_below(mid[3 - m]);
      win[H + 4 + m] = from_lane_above(mid[m]);
    }
    float out[4];
#pragma unroll
    for (int ox = 0; ox < 4; ox++) {
      float a = w.kx[0] * win[ox];
#pragma unroll
      for (int j = 1; j < K; j++) a = fmaf(w.kx[j], win[ox + j], a);
      out[ox] = a;
    }
    return v4f{out[0], out[1], out[2], out[3]};
  }
};

// batches of bilinear-sampled frames share their footprint records through LDS (any coordinate
// source, K = 3..9: the C3 chain - homography + separable 9+9 - evaluates its double
// coordinates once per four frames)
template <typename Src, int K> struct sep_shared : std::false_type {};
template <typename ST, typename Coord, int K> struct sep_shared<SampleRowSrc<ST, kLinear, Coord>, K> {
  static constexpr bool value = IPA_PIPE != 0 && IPA_PIPE_SHARED != 0 &&
                                (std::is_same<ST, float>::value || std::is_same<ST, uint16_t>::value ||
                                 std::is_same<ST, uint8_t>::value);
};

template <typename Src> struct sep_shares_maps : std::false_type {};
template <typename ST, int I, typename Coord>
struct sep_shares_maps<SampleRowSrc<ST, I, Coord>> {
  static constexpr bool value = coord_is_table<Coord>::value ||
                                (I == kLinear && IPA_PIPE != 0 && IPA_PIPE_SHARED != 0);
};

// DT: the destination's element type - float32, or (K = 1, integer frames with cv2's own arithmetic on the shared-record
// loop only: wave_pipe.hpp CV16) uint16 / uint8; the host launches that form only where every strip takes that loop
template <typename Src, int K, typename DT = float>
__global__ void __launch_bounds__(256)
wave_sep_kernel(WaveParams p, Src src, SepTaps<K> w, float xcval) {
  constexpr bool kCv16 = std::is_same<DT, uint16_t>::value || std::is_same<DT, uint8_t>::value;   // integer results
  static_assert(std::is_same<DT, float>::value || (kCv16 && K == 1), "float32 results, or the integer strip remaps");
  constexpr int H = K / 2, D = sep_depth<Src, K>::value, HL = sep_geom<K>::HL, OW = sep_geom<K>::OW;
  constexpr bool kRegs = std::is_same<Src, LoadRowSrc>::value;
  __shared__ __attribute__((aligned(16))) float xpose[kRegs ? 1 : 4 * kRowStride * D];
  static_assert(H <= 4 * HL, "the halo lanes hold the x pass's neighbours");
  const int lane = threadIdx.x & 63;
  unsigned b = xcd_swizzle(blockIdx.x, gridDim.x), frame = blockIdx.y;
  if (p.frame_major) {   // dispatch orders of wave_stencil_kernel
    frame = b / p.frame_major;
    b -= frame * p.frame_major;
  } else if (p.frames_inner) {
    frame = b % (unsigned)p.frames_inner;
    b /= (unsigned)p.frames_inner;
  }
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar
  unsigned sid = b * 4 + wave;
  if (p.frames_wg) {  // the waves of a workgroup are 4 frames of one strip (WaveParams::frames_wg)
    const unsigned groups = (unsigned)p.frames_inner / 4u;
    b = xcd_swizzle(blockIdx.x, gridDim.x);
    if (p.group_chunk) {   // (wave_stencil.hpp: the frame groups a chunk at a time)
      const unsigned gc = (unsigned)p.group_chunk, per = gc * p.strips;
      const unsigned chunk = b / per, r = b - chunk * per;
      frame = (chunk * gc + r % gc) * 4u + wave;
      sid = r / gc;
    } else {
      frame = (b % groups) * 4u + wave;
      sid = b / groups;
    }
  }
  float* xp = xpose + (kRegs ? 0 : wave * kRowStride * D);
  constexpr bool kShared = sep_shared<Src, K>::value;
  __shared__ __attribute__((aligned(16))) float mapring[kShared ? 2 * 4 * ring_row<false>::value : 4];
  static_assert(!kShared || IPA_WPB == 4, "the separable kernel runs 4 waves per workgroup");
  if (sid >= p.strips) return;
  const int syi = (int)(sid / (unsigned)p.strips_x), sxi = (int)sid - syi * p.strips_x;
  src.set_frame(frame);
  const int xs = sxi * OW - 4 * HL;
  Cols c;
  c.xs = xs;
  c.xo = xs + lane * 4;
  int y0, nrows;
  if (!wave_strip_rows(p, syi, y0, nrows)) return;
  const bool writer = lane >= HL && lane < 64 - HL && c.xo < p.dw;
  // (uint16 results: the same pointer arithmetic in their elements; the loop below reinterprets it)
  float* dst = reinterpret_cast<float*>(reinterpret_cast<DT*>(p.dst) + (long)frame * p.dst_frame_elems);
  const int rows_touched = ((nrows + K - 1 + D - 1) / D) * D;
  const bool fast = !p.no_pipe && src.vectors_ok() && p.vec_out && xs >= 0 && xs + 256 <= p.dw &&
                    y0 - H >= 0 && y0 - H + rows_touched <= p.dh;
  if (fast) {
#pragma unroll
    for (int k = 0; k < 4; k++) c.uu[k] = c.xo + k;
    if constexpr (kShared) {
      if (p.frames_wg) {   // every wave of the workgroup: the same strip of another frame
        SepFilter<K> filt(w, xcval);
        if constexpr (kCv16) {
          wave_run_strip_shared<K, 1, false, false, SepFilter<K>, typename Src::sample_type, typename Src::coord_type, true>(
              p, src, filt, xp, mapring, wave, c, y0, nrows, writer, dst);
        } else {
          if (src.q5) wave_run_strip_shared<K, 1, false, false>(p, src, filt, xp, mapring, wave, c, y0, nrows, writer, dst);
          else wave_run_strip_shared<K, 0, false, false>(p, src, filt, xp, mapring, wave, c, y0, nrows, writer, dst);
        }
        return;
      }
    }
    if constexpr (kCv16) return;   // (never: the host launches this form only where every strip takes the loop above)
    else if constexpr (!kRegs) {
      if (src.q5) wave_sep_strip<true, Src, K, 1>(p, src, w, xp, c, y0, nrows, writer, dst, xcval);
      else wave_sep_strip<true, Src, K, 0>(p, src, w, xp, c, y0, nrows, writer, dst, xcval);
    } else {
      wave_sep_strip<true, Src, K>(p, src, w, xp, c, y0, nrows, writer, dst, xcval);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      c.uu[k] = resolve_idx(c.xo + k, p.dw, p.cbx);
      c.uq[k] = resolve_idx(xs + lane + 64 * k, p.dw, p.cbx);
    }
    if constexpr (kShared) {
      if (p.frames_wg && !p.no_pipe && src.vectors_ok() && p.vec_out && (p.dw & 3) == 0 && IPA_PIPE_EDGE) {
        SepFilter<K> filt(w, xcval);
        if constexpr (kCv16) {
          wave_run_strip_shared<K, 1, true, false, SepFilter<K>, typename Src::sample_type, typename Src::coord_type, true>(
              p, src, filt, xp, mapring, wave, c, y0, nrows, writer, dst);
        } else {
          if (src.q5) wave_run_strip_shared<K, 1, true, false>(p, src, filt, xp, mapring, wave, c, y0, nrows, writer, dst);
          else wave_run_strip_shared<K, 0, true, false>(p, src, filt, xp, mapring, wave, c, y0, nrows, writer, dst);
        }
        return;
      }
    }
    if constexpr (!kCv16) wave_sep_strip<false, Src, K>(p, src, w, xp, c, y0, nrows, writer, dst, xcval);
  }
}

template <typename Src, int K, typename DT = float>
static void launch_sep(ipa_ctx* ctx, WaveParams p, const Src& src, const double* ky,
                       const double* kx, int n_frames, float xcval) {
  SepTaps<K> w;
  for (int i = 0; i < K; i++) {
    w.ky[i] = (float)ky[i];
    w.kx[i] = (float)kx[i];
  }
  p.strips_x = (p.dw + sep_geom<K>::OW - 1) / sep_geom<K>::OW;
  // (the tall strips of the shared-record loop only where that loop runs, see fused_strip_piped)
  const bool shared_run = sep_shared<Src, K>::value && sep_shares_maps<Src>::value &&
                          ctx->tune.frames_wg != 0 && ctx->tune.frames_inner != 0 && n_frames % 4 == 0;
  // plain rows: the short strips of the plain dense filters (64 x 4K, 9 + 9 taps: 72 rows 0.906, 24 rows 0.882 ms)
  p.strip_h = wave_strip_height(ctx, p.dh, p.dw, n_frames, K, false,
                                shared_run ? 2 : (std::is_same<Src, LoadRowSrc>::value ? 1 : 0), p.strips_x);
  p.strips = (unsigned)p.strips_x * (unsigned)((p.dh + p.strip_h - 1) / p.strip_h);
  // plain rows (LoadRowSrc): frames share nothing - frame after frame, every XCD streaming through frames of its
  // own (knob frame_major, as the dense plain filters since round 4)
  dim3 grid = wave_grid(ctx, p, n_frames, 4, true, sep_shares_maps<Src>::value,
                        std::is_same<Src, LoadRowSrc>::value, -K), block(256);
  hipLaunchKernelGGL((wave_sep_kernel<Src, K, DT>), grid, block, 0, ctx->stream, p, src, w, xcval);
}

}  // namespace ipa
