// wave_pair.hpp - the fused bilinear remap -> K x K filter (wave_stencil.hpp) with ONE wave
// working on the same strip of TWO frames of a batch: the map rows are loaded once, footprint
// offsets, validity and bilinear fractions are computed once and used for both frames' gathers
// (per 256-px row and frame: 16 dword gathers + 4 map dwords + 1 store instead of 16 + 8 + 1 on
// a kernel bound by the CU's vector-memory path).  Interior strips only; rim strips and the odd
// last frame run the per-frame strip code, frame after frame.
// Same arithmetic and order as wave_stencil_kernel: identical bits.
// Reference call chain: camera/LensDistortion.py:323-326 + filters/maskedConvolve.py:24-43.
#pragma once
#include "wave_stencil.hpp"

#ifndef IPA_PAIR_DEPTH
#define IPA_PAIR_DEPTH 2
#endif

namespace ipa {

template <int K, int QM>
__device__ __forceinline__ void wave_run_strip_pair(
    const WaveParams& p, const SampleRowSrc<float, kLinear, MapCoord>& src, const SrcView& s0,
    const SrcView& s1, const Weights<float, K * K>& wts, float* xp0, float* xp1, const Cols& c,
    int y0, int nrows, bool writer, float* dst0, float* dst1) {
  using G = wave_geom<K>;
  constexpr int D = IPA_PAIR_DEPTH;
  const int T = nrows + K - 1;
  const unsigned lane = threadIdx.x & 63u;
  unsigned lane4_opaque = 4u * lane;
  asm volatile("" : "+v"(lane4_opaque));
  v2f acc[2][K][2];
#pragma unroll 1
  for (int tb = 0; tb < T; tb += D) {
    int vv[D];
#pragma unroll
    for (int d = 0; d < D; d++) vv[d] = y0 - G::H + tb + d;
    // map rows of the chunk, once
    float sx[D][4], sy[D][4];
#pragma unroll
    for (int d = 0; d < D; d++) {
      const long o = (long)vv[d] * src.coord.pitch + c.xs;  // scalar
      const float* rx = src.coord.mx + o;
      const float* ry = src.coord.my + o;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        sx[d][k] = rx[lane + 64u * k];
        sy[d][k] = ry[lane + 64u * k];
      }
    }
    // footprints once, gathers per frame
    BatchTaps<float, kLinear, 4> t0[D], t1[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
      int e[4];
      batch_footprint_linear<4, QM>(s0, sx[d], sy[d], t0[d].tx, t0[d].ty, e, t0[d].interior);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        t1[d].tx[k] = t0[d].tx[k];
        t1[d].ty[k] = t0[d].ty[k];
      }
      t1[d].interior = t0[d].interior;
      batch_loads_linear<4>(s0, e, t0[d]);
      batch_loads_linear<4>(s1, e, t1[d]);
    }
    // blend, rows to the wave's LDS rows in natural pixel order
#pragma unroll
    for (int f = 0; f < 2; f++) {
      const SrcView& s = f ? s1 : s0;
      float* xp = f ? xp1 : xp0;
#pragma unroll
      for (int d = 0; d < D; d++) {
        const BatchTaps<float, kLinear, 4>& t = f ? t1[d] : t0[d];
        float cur[4];
#pragma unroll
        for (int k = 0; k < 4; k++) cur[k] = batch_blend_one<float, kLinear, 4>(s, t, k);
        if (t.interior != 0xfu) {
#pragma unroll
          for (int k = 0; k < 4; k++)
            if (!((t.interior >> k) & 1u))
              cur[k] = sample<float, kLinear, float>(s, sx[d][k], sy[d][k], src.cval);
        }
        float* row = xp + d * kRowStride + kRowPad;
#pragma unroll
        for (int k = 0; k < 4; k++) row[64u * k + lane] = cur[k];
      }
    }
    __builtin_amdgcn_wave_barrier();

    static_for<0, D>([&](auto Dd) {
      constexpr int d = decltype(Dd)::value;
      const int t = tb + d;
      static_for<0, 2>([&](auto Ff) {
        constexpr int f = decltype(Ff)::value;
        float* xp = f ? xp1 : xp0;
        const float* wp = xp + d * kRowStride + kRowPad - G::H + 4u * lane;
        const float* wq = xp + d * kRowStride + kRowPad - G::H + lane4_opaque;
        v2f pair[K + 2];
#pragma unroll
        for (int m = 0; m < K + 2; m++)
          pair[m] = (m & 1) ? v2f{wq[m], wq[m + 1]} : v2f{wp[m], wp[m + 1]};
        static_for<0, K>([&](auto Ii) {
          constexpr int i = K - 1 - decltype(Ii)::value;
#pragma unroll
          for (int j = 0; j < K; j++) {
            const float w = wts.w[i * K + j];
            const v2f w2 = v2f{w, w};
#pragma unroll
            for (int h = 0; h < 2; h++) {
              if constexpr (i == 0) {
                acc[f][0][h] = j == 0 ? w2 * pair[2 * h]
                                      : __builtin_elementwise_fma(w2, pair[j + 2 * h], acc[f][0][h]);
              } else {
                acc[f][i][h] = __builtin_elementwise_fma(w2, pair[j + 2 * h],
                                                         j == 0 ? acc[f][i - 1][h] : acc[f][i][h]);
              }
            }
          }
        });
        const int o = t - (K - 1);
        if (o >= 0 && o < nrows && writer) {
          float* rows_ = (f ? dst1 : dst0) + ((long)(y0 + o) * p.dpitch + c.xs);  // scalar base
          __builtin_nontemporal_store(acc[f][K - 1][0].x, rows_ + 4u * lane);
          __builtin_nontemporal_store(acc[f][K - 1][0].y, rows_ + 4u * lane + 1);
          __builtin_nontemporal_store(acc[f][K - 1][1].x, rows_ + 4u * lane + 2);
          __builtin_nontemporal_store(acc[f][K - 1][1].y, rows_ + 4u * lane + 3);
        }
      });
    });
    __builtin_amdgcn_wave_barrier();
  }
}

template <int K>
__global__ void __launch_bounds__(64 * IPA_WPB)
wave_pair_kernel(WaveParams p, SampleRowSrc<float, kLinear, MapCoord> src,
                 Weights<float, K * K> wts, int n_frames) {
  using Src = SampleRowSrc<float, kLinear, MapCoord>;
  using G = wave_geom<K>;
  constexpr int D = IPA_PAIR_DEPTH;
  const int lane = threadIdx.x & 63;
  // frame pairs of one strip block are neighbours in the XCD-contiguous order
  const unsigned pairs = ((unsigned)n_frames + 1u) / 2u;
  unsigned b = xcd_swizzle(blockIdx.x, gridDim.x);
  const unsigned fp = b % pairs;
  b /= pairs;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned sid = b * IPA_WPB + wave;
  constexpr int kXp = kRowStride * D;
  __shared__ __attribute__((aligned(16))) float xpose[IPA_WPB * 2 * kXp];
  float* xp0 = xpose + wave * 2 * kXp;
  float* xp1 = xp0 + kXp;
  if (sid >= p.strips) return;
  const int syi = (int)(sid / (unsigned)p.strips_x), sxi = (int)sid - syi * p.strips_x;
  const unsigned f0 = 2u * fp;
  const bool two = f0 + 1u < (unsigned)n_frames;

  const int xs = sxi * G::OW - 4 * G::HL;
  Cols c;
  c.xs = xs;
  c.xo = xs + lane * 4;
  const int y0 = syi * p.strip_h;
  const int nrows = p.dh - y0 < p.strip_h ? p.dh - y0 : p.strip_h;
  const bool writer = lane >= G::HL && lane < 64 - G::HL && c.xo < p.dw;
  float* dst0 = reinterpret_cast<float*>(p.dst) + (long)f0 * p.dst_frame_elems;
  float* dst1 = dst0 + p.dst_frame_elems;
  const int rows_touched = ((nrows + K - 1 + D - 1) / D) * D;
  const bool fast = src.vectors_ok() && p.vec_out && xs >= 0 && xs + 256 <= p.dw &&
                    y0 - G::H >= 0 && y0 - G::H + rows_touched <= p.dh;
  if (fast && two) {
#pragma unroll
    for (int k = 0; k < 4; k++) c.uu[k] = c.xo + k;
    src.set_frame(f0);
    const SrcView s0 = src.s;
    src.set_frame(f0 + 1u);
    const SrcView s1 = src.s;
    if (src.q5)
      wave_run_strip_pair<K, 1>(p, src, s0, s1, wts, xp0, xp1, c, y0, nrows, writer, dst0, dst1);
    else
      wave_run_strip_pair<K, 0>(p, src, s0, s1, wts, xp0, xp1, c, y0, nrows, writer, dst0, dst1);
    return;
  }
  // rim strips / the odd last frame: the per-frame strip code, frame after frame
#pragma unroll 1
  for (unsigned f = f0; f < f0 + (two ? 2u : 1u); f++) {
    src.set_frame(f);
    float* dst = reinterpret_cast<float*>(p.dst) + (long)f * p.dst_frame_elems;
    if (fast) {
#pragma unroll
      for (int k = 0; k < 4; k++) c.uu[k] = c.xo + k;
      wave_run_strip<true, Src, K, -1, false>(p, src, wts, xp0, c, y0, nrows, writer, dst, nullptr);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        c.uu[k] = resolve_idx(c.xo + k, p.dw, p.cbx);
        c.uq[k] = resolve_idx(xs + lane + 64 * k, p.dw, p.cbx);
      }
      wave_run_strip<false, Src, K, -1, false>(p, src, wts, xp0, c, y0, nrows, writer, dst, nullptr);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace ipa
