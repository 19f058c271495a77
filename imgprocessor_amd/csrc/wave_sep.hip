// wave_sep.hip — separable K+K correlation (scipy.ndimage.gaussian_filter order: axis 0
// then axis 1, intermediate rounded to float32) on the wave-marching skeleton of
// wave_stencil.hpp, float32, K = 3, 5, 7, 9 taps on both axes.
//
// A wave64 owns a 256-px-wide strip.  Each arriving input row is scattered into the K
// running y-sums it belongs to (K x 4 registers per lane, shifted inside the fma
// chain); the y-row that completes is already the float32 intermediate scipy
// stores between its two passes; its x pass needs K/2 neighbours per side,
// taken from the adjacent lanes with DPP wave shifts, and the result is stored
// as one float4.  2K fma per pixel instead of K*K, one pass over HBM (8 B/px),
// no LDS, no barriers.  Reference call sites: filters/standardDeviation.py:23,
// filters/fastFilter.py:42, camera/flatField/flatField.py:47.
#include "common.hpp"
#include "wave_stencil.hpp"

namespace ipa {

template <int K> struct SepTaps {
  float ky[K], kx[K];
};

template <bool FAST, int K>
__device__ __forceinline__ void wave_sep_strip(const WaveParams& p, const LoadRowSrc& src,
                                               const SepTaps<K>& w, const Cols& c, int y0,
                                               int nrows, bool writer, float* dst, float xcval) {
  constexpr int H = K / 2;
  constexpr int D = 4;
  const int T = nrows + K - 1;
  float acc[K][4];  // acc[i] = y-sum of intermediate row (t - i)
#pragma unroll 1
  for (int tb = 0; tb < T; tb += D) {
    int vv[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
      if constexpr (FAST) vv[d] = y0 - H + tb + d;
      else vv[d] = resolve_idx(y0 - H + tb + d, p.dh, p.cby);
    }
    LoadRowSrc::Chunk<D> ch;
    src.template load_chunk<FAST, D>(c, vv, ch);

    static_for<0, D>([&](auto Dd) {
      constexpr int d = decltype(Dd)::value;
      const int t = tb + d;
      float cur[4];
#pragma unroll
      for (int k = 0; k < 4; k++) cur[k] = ch.v[d][k];

      // y pass: the arriving row feeds K intermediate rows
      static_for<0, K>([&](auto Ii) {
        constexpr int i = K - 1 - decltype(Ii)::value;
#pragma unroll
        for (int ox = 0; ox < 4; ox++) {
          if constexpr (i == 0) acc[0][ox] = w.ky[0] * cur[ox];
          else acc[i][ox] = fmaf(w.ky[i], cur[ox], acc[i - 1][ox]);
        }
      });

      const int o = t - (K - 1);
      if (o >= 0 && o < nrows) {  // wave-uniform: intermediate row o is complete
        float mid[4];
#pragma unroll
        for (int k = 0; k < 4; k++) mid[k] = acc[K - 1][k];
        if constexpr (!FAST) {
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
  }
}

template <int K>
__global__ void __launch_bounds__(256)
wave_sep_kernel(WaveParams p, LoadRowSrc src, SepTaps<K> w, float xcval) {
  constexpr int H = K / 2, D = 4, OW = 256 - 8;
  static_assert(H <= 4, "one halo lane per side");
  const int lane = threadIdx.x & 63;
  unsigned b = xcd_swizzle(blockIdx.x, gridDim.x), frame = blockIdx.y;
  if (p.frames_inner) {  // dispatch order of wave_stencil_kernel
    frame = b % (unsigned)p.frames_inner;
    b /= (unsigned)p.frames_inner;
  }
  const unsigned sid = b * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar
  if (sid >= p.strips) return;
  const int syi = (int)(sid / (unsigned)p.strips_x), sxi = (int)sid - syi * p.strips_x;
  src.set_frame(frame);
  const int xs = sxi * OW - 4;
  Cols c;
  c.xs = xs;
  c.xo = xs + lane * 4;
  const int y0 = syi * p.strip_h;
  const int nrows = p.dh - y0 < p.strip_h ? p.dh - y0 : p.strip_h;
  const bool writer = lane >= 1 && lane < 63 && c.xo < p.dw;
  float* dst = reinterpret_cast<float*>(p.dst) + (long)frame * p.dst_frame_elems;
  const int rows_touched = ((nrows + K - 1 + D - 1) / D) * D;
  const bool fast = src.vectors_ok() && p.vec_out && xs >= 0 && xs + 256 <= p.dw && y0 - H >= 0 &&
                    y0 - H + rows_touched <= p.dh;
  if (fast) {
#pragma unroll
    for (int k = 0; k < 4; k++) c.uu[k] = c.xo + k;
    wave_sep_strip<true, K>(p, src, w, c, y0, nrows, writer, dst, xcval);
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) c.uu[k] = resolve_idx(c.xo + k, p.dw, p.cbx);
    wave_sep_strip<false, K>(p, src, w, c, y0, nrows, writer, dst, xcval);
  }
}

template <int K>
static void launch_sep(ipa_ctx* ctx, WaveParams p, const LoadRowSrc& src, const double* ky,
                       const double* kx, int n_frames, float xcval) {
  SepTaps<K> w;
  for (int i = 0; i < K; i++) {
    w.ky[i] = (float)ky[i];
    w.kx[i] = (float)kx[i];
  }
  p.strips_x = (p.dw + 247) / 248;
  p.strip_h = wave_strip_height(p.dh, p.dw, n_frames, K);
  p.strips = (unsigned)p.strips_x * (unsigned)((p.dh + p.strip_h - 1) / p.strip_h);
  dim3 grid = wave_grid(p, n_frames, 4, true), block(256);
  hipLaunchKernelGGL((wave_sep_kernel<K>), grid, block, 0, ctx->stream, p, src, w, xcval);
}

}  // namespace ipa

// returns 1 when the shape is not covered (caller falls back to the LDS kernel)
int ipa_wave_sep_launch(ipa_ctx* ctx, const ipa::WaveParams& p, const ipa::LoadRowSrc& src,
                        const double* ky, int nky, const double* kx, int nkx, int n_frames,
                        float xcval) {
  using namespace ipa;
  if (nky != nkx) return 1;
  switch (nky) {
    case 3: launch_sep<3>(ctx, p, src, ky, kx, n_frames, xcval); return 0;
    case 5: launch_sep<5>(ctx, p, src, ky, kx, n_frames, xcval); return 0;
    case 7: launch_sep<7>(ctx, p, src, ky, kx, n_frames, xcval); return 0;
    case 9: launch_sep<9>(ctx, p, src, ky, kx, n_frames, xcval); return 0;
  }
  return 1;
}
