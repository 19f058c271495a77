// wave_split.hpp - the fused bilinear remap -> K x K filter (wave_stencil.hpp) with the two halves
// of a strip's work on TWO waves of one workgroup:
//   wave 0 (sampler): map rows, footprints, tap gathers, blend -> rows in LDS.  It only LOADS, so
//     its waits are counted (a wave that also stores can only wait for a load with vmcnt(0),
//     DESIGN.md section 5) and the next chunk's map rows are requested before this chunk's taps
//     are waited for;
//   wave 1 (filter): row windows from LDS, the K x K running sums, stores.  It never waits for
//     memory.
// Two LDS row buffers, one workgroup barrier per chunk of D rows.  Interior strips only; the rim
// strips are left to wave_stencil_kernel (WaveParams::rim_only).
// Same arithmetic and order as wave_stencil_kernel: identical bits.
// Reference call chain: camera/LensDistortion.py:323-326 + filters/maskedConvolve.py:24-43.
#pragma once
#include "wave_stencil.hpp"

namespace ipa {

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for vmcnt(0), i.e.
// for the sampler's prefetched map rows and for the filter's stores
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int K, int QM>
__device__ __forceinline__ void split_sample(const WaveParams& p,
                                             const SampleRowSrc<float, kLinear, MapCoord>& src,
                                             float* xp2, const Cols& c, int y0, int nrows) {
  using Src = SampleRowSrc<float, kLinear, MapCoord>;
  using G = wave_geom<K>;
  constexpr int D = Src::template depth<K>::value;
  constexpr int kXp = kRowStride * D;
  const int T = nrows + K - 1;
  const unsigned lane = threadIdx.x & 63u;
  float nx[D][4], ny[D][4];
  auto request_maps = [&](int tb) {
#pragma unroll
    for (int d = 0; d < D; d++) {
      const long o = (long)(y0 - G::H + tb + d) * src.coord.pitch + c.xs;  // scalar
      const float* rx = src.coord.mx + o;
      const float* ry = src.coord.my + o;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        nx[d][k] = rx[lane + 64u * k];
        ny[d][k] = ry[lane + 64u * k];
      }
    }
  };
  request_maps(0);
  int buf = 0;
#pragma unroll 1
  for (int tb = 0; tb < T; tb += D, buf ^= 1) {
    int vv[D];
#pragma unroll
    for (int d = 0; d < D; d++) vv[d] = y0 - G::H + tb + d;
    typename Src::template Chunk<D> ch;
#pragma unroll
    for (int d = 0; d < D; d++) batch_issue<float, kLinear, 4, QM>(src.s, nx[d], ny[d], ch.t[d]);
    // the next chunk's map rows behind this chunk's gathers (the last chunk asks for nothing)
    if (tb + D < T) request_maps(tb + D);
    src.template stage_rows<true, D>(c, vv, ch, xp2 + buf * kXp);
    lds_barrier();
  }
}

template <int K>
__device__ __forceinline__ void split_filter(const WaveParams& p, const Weights<float, K * K>& wts,
                                             const float* xp2, const Cols& c, int y0, int nrows,
                                             bool writer, float* dst) {
  using Src = SampleRowSrc<float, kLinear, MapCoord>;
  using G = wave_geom<K>;
  constexpr int D = Src::template depth<K>::value;
  constexpr int kXp = kRowStride * D;
  const int T = nrows + K - 1;
  const unsigned lane = threadIdx.x & 63u;
  unsigned lane4_opaque = 4u * lane;
  asm volatile("" : "+v"(lane4_opaque));
  v2f acc[K][2];
  int buf = 0;
#pragma unroll 1
  for (int tb = 0; tb < T; tb += D, buf ^= 1) {
    lds_barrier();
    const float* xp = xp2 + buf * kXp;
    static_for<0, D>([&](auto Dd) {
      constexpr int d = decltype(Dd)::value;
      const int t = tb + d;
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
              acc[0][h] = j == 0 ? w2 * pair[2 * h]
                                 : __builtin_elementwise_fma(w2, pair[j + 2 * h], acc[0][h]);
            } else {
              acc[i][h] = __builtin_elementwise_fma(w2, pair[j + 2 * h],
                                                    j == 0 ? acc[i - 1][h] : acc[i][h]);
            }
          }
        }
      });
      const int o = t - (K - 1);
      if (o >= 0 && o < nrows && writer) {
        float* rows_ = dst + ((long)(y0 + o) * p.dpitch + c.xs);  // scalar base
        __builtin_nontemporal_store(acc[K - 1][0].x, rows_ + 4u * lane);
        __builtin_nontemporal_store(acc[K - 1][0].y, rows_ + 4u * lane + 1);
        __builtin_nontemporal_store(acc[K - 1][1].x, rows_ + 4u * lane + 2);
        __builtin_nontemporal_store(acc[K - 1][1].y, rows_ + 4u * lane + 3);
      }
    });
  }
}

template <int K>
#ifndef IPA_SPLIT_MIN_WAVES
#define IPA_SPLIT_MIN_WAVES 1
#endif
__global__ void __launch_bounds__(128, IPA_SPLIT_MIN_WAVES)
wave_split_kernel(WaveParams p, SampleRowSrc<float, kLinear, MapCoord> src,
                  Weights<float, K * K> wts) {
  using Src = SampleRowSrc<float, kLinear, MapCoord>;
  using G = wave_geom<K>;
  constexpr int D = Src::template depth<K>::value;
  constexpr int kXp = kRowStride * D;
  __shared__ __attribute__((aligned(16))) float xpose[2 * kXp];
  const int lane = threadIdx.x & 63;
  // one workgroup = one strip of one frame; the frames of a strip are neighbours in the
  // XCD-contiguous order (p.frames_inner = n_frames)
  unsigned b = xcd_swizzle(blockIdx.x, gridDim.x);
  const unsigned frame = b % (unsigned)p.frames_inner;
  const unsigned sid = b / (unsigned)p.frames_inner;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (p.skip && p.skip[sid]) return;
  const int syi = (int)(sid / (unsigned)p.strips_x), sxi = (int)sid - syi * p.strips_x;
  src.set_frame(frame);
  const int xs = sxi * G::OW - 4 * G::HL;
  Cols c;
  c.xs = xs;
  c.xo = xs + lane * 4;
  const int y0 = syi * p.strip_h;
  const int nrows = p.dh - y0 < p.strip_h ? p.dh - y0 : p.strip_h;
  const bool writer = lane >= G::HL && lane < 64 - G::HL && c.xo < p.dw;
  float* dst = reinterpret_cast<float*>(p.dst) + (long)frame * p.dst_frame_elems;
  const int rows_touched = ((nrows + K - 1 + D - 1) / D) * D;
  const bool fast = src.vectors_ok() && p.vec_out && xs >= 0 && xs + 256 <= p.dw &&
                    y0 - G::H >= 0 && y0 - G::H + rows_touched <= p.dh;
  if (fast) {
#pragma unroll
    for (int k = 0; k < 4; k++) c.uu[k] = c.xo + k;
    if (wave == 0) {
      if (src.q5) split_sample<K, 1>(p, src, xpose, c, y0, nrows);
      else split_sample<K, 0>(p, src, xpose, c, y0, nrows);
    } else {
      split_filter<K>(p, wts, xpose, c, y0, nrows, writer, dst);
    }
  }
  // rim strips: wave_stencil_kernel with p.rim_only (their border code would set this kernel's
  // register allocation)
}

}  // namespace ipa
