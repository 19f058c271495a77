// fused_impl.hpp — remap -> K x K filter in ONE kernel (the headline chain:
// LensDistortion.correct / PerspectiveCorrection.correct followed by a dense
// filter; in-tree archetype camera/lens/estimateSystematicErrorLensCorrection.py:199-207).
//
// A workgroup owns a 128 x 32 output tile.  Phase 1 samples the remapped image
// for the tile plus its K/2 halo straight into LDS (halo positions outside
// the remapped image are resolved with the FILTER's border mode first, i.e.
// exactly what filtering the materialised remap result would see).  Phase 2 is
// conv_from_lds.  The intermediate image never exists in HBM: 16 B/px map-
// based, 8 B/px analytic, instead of 24 / 16 for two launches.
//
// This header is compiled once per K (fused_k*.hip define IPA_FUSED_K) so the
// translation units build in parallel.
#pragma once

#include "common.hpp"
#include "conv_tile.hpp"
#include "sampler.hpp"

namespace ipa {

struct FusedParams {
  const char* src;
  char* dst;
  long src_frame_bytes, dst_frame_elems;
  unsigned src_bytes;
  int sh, sw, spitch;
  int dh, dw;
  long dpitch;
  int border, q5;
  float cubic_a;
  const float* lanczos;
  double cval;
  int cbx, cby;      // filter border mode per axis, applied on the remapped image
  double conv_cval;  // filter border value (IPA_BORDER_CONSTANT)
  unsigned tiles_x, tiles;
  int vec_out;
};

// host-side description of one fused call (coordinate source by kind)
struct FusedCall {
  FusedParams p;
  int coord_kind;  // 0 map, 1 undistort, 2 homography
  MapCoord map;
  UndistortCoord und;
  HomographyCoord hom;
  int src_dt, dst_dt, interp_base, n_frames;
  const double* kernel;
};

template <typename ST, int INTERP, typename Coord, int K>
__global__ void __launch_bounds__(256)
fused_kernel(FusedParams p, Coord coord, Weights<typename compute_of<ST>::type, K * K> wts) {
  using CT = typename compute_of<ST>::type;
  using G = conv_geom<K>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  CT* tile = reinterpret_cast<CT*>(smem);

  unsigned t = xcd_swizzle(blockIdx.x, p.tiles);
  unsigned tyi = t / p.tiles_x, txi = t - tyi * p.tiles_x;
  const int x0 = (int)txi * kTileW, y0 = (int)tyi * kTileH;
  const unsigned frame = blockIdx.y;
  const int tid = threadIdx.y * 32 + threadIdx.x;

  SrcView s;
  s.rsrc = make_rsrc(p.src + (long)frame * p.src_frame_bytes, p.src_bytes);
  s.h = p.sh; s.w = p.sw; s.pitch = p.spitch;
  s.border = p.border; s.q5 = p.q5; s.cubic_a = p.cubic_a; s.lanczos = p.lanczos;
  const CT cval = (CT)p.cval, ccval = (CT)p.conv_cval;

  // ---- phase 1: sample tile + halo into LDS (consecutive lanes -> consecutive x)
  constexpr int NWC = kTileW + K - 1;
  constexpr int ROWS = kTileH + K - 1;
  for (int idx = tid; idx < NWC * ROWS; idx += 256) {
    int lr = idx / NWC, c = idx - lr * NWC;
    int uu = resolve_idx(x0 - K / 2 + c, p.dw, p.cbx);
    int vv = resolve_idx(y0 - K / 2 + lr, p.dh, p.cby);
    CT val = ccval;
    if (uu >= 0 && vv >= 0) {
      typename Coord::coord_t sx, sy;
      coord.get(uu, vv, sx, sy);
      val = sample<ST, INTERP>(s, sx, sy, cval);
    }
    tile[lr * G::LW + G::OFF + c] = val;
  }
  __syncthreads();

  // ---- phase 2: K x K correlation out of LDS
  CT acc[4][4];
  conv_from_lds<CT, K, K>(tile, threadIdx.x, threadIdx.y, wts, acc);

  int ox = x0 + threadIdx.x * 4;
  CT* dst = reinterpret_cast<CT*>(p.dst) + (long)frame * p.dst_frame_elems;
#pragma unroll
  for (int oy = 0; oy < 4; oy++) {
    int y = y0 + threadIdx.y * 4 + oy;
    if (y >= p.dh || ox >= p.dw) continue;
    int n = p.dw - ox < 4 ? p.dw - ox : 4;
    CT* row = dst + (long)y * p.dpitch + ox;
    if (p.vec_out && n == 4) {
      if constexpr (sizeof(CT) == 4) {
        *reinterpret_cast<float4*>(row) = float4{acc[oy][0], acc[oy][1], acc[oy][2], acc[oy][3]};
      } else {
        reinterpret_cast<double2*>(row)[0] = double2{acc[oy][0], acc[oy][1]};
        reinterpret_cast<double2*>(row)[1] = double2{acc[oy][2], acc[oy][3]};
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (k < n) row[k] = acc[oy][k];
    }
  }
}

template <typename ST, int INTERP, typename Coord, int K>
static void fused_launch_one(ipa_ctx* ctx, const FusedCall& f, const Coord& c) {
  using CT = typename compute_of<ST>::type;
  using G = conv_geom<K>;
  Weights<CT, K * K> w;
  for (int i = 0; i < K * K; i++) w.w[i] = (CT)f.kernel[i];
  size_t lds = (size_t)lds_rows<K>() * G::LW * sizeof(CT);
  dim3 grid(f.p.tiles, (unsigned)f.n_frames), block(32, 8);
  hipLaunchKernelGGL((fused_kernel<ST, INTERP, Coord, K>), grid, block, lds, ctx->stream, f.p, c, w);
}

template <typename ST, typename Coord, int K>
static int fused_launch_interp(ipa_ctx* ctx, const FusedCall& f, const Coord& c) {
  switch (f.interp_base) {
    case IPA_INTER_LINEAR: fused_launch_one<ST, kLinear, Coord, K>(ctx, f, c); break;
    case IPA_INTER_CUBIC_CV:
    case IPA_INTER_CUBIC_KEYS: fused_launch_one<ST, kCubic, Coord, K>(ctx, f, c); break;
    default:
      IPA_UNSUPPORTED(ctx, "fused remap+filter supports INTER_LINEAR and the two bicubics");
  }
  return IPA_OK;
}

template <typename ST, int K> static int fused_launch_coord(ipa_ctx* ctx, const FusedCall& f) {
  switch (f.coord_kind) {
    case 0: return fused_launch_interp<ST, MapCoord, K>(ctx, f, f.map);
    case 1: return fused_launch_interp<ST, UndistortCoord, K>(ctx, f, f.und);
    default: return fused_launch_interp<ST, HomographyCoord, K>(ctx, f, f.hom);
  }
}

template <int K> static int fused_launch_k(ipa_ctx* ctx, const FusedCall& f) {
  if (f.src_dt == IPA_F32 && f.dst_dt == IPA_F32) return fused_launch_coord<float, K>(ctx, f);
  if (f.src_dt == IPA_U16 && f.dst_dt == IPA_F32) return fused_launch_coord<uint16_t, K>(ctx, f);
  IPA_UNSUPPORTED(ctx, "fused remap+filter: src dtype %d -> dst dtype %d not supported "
                       "(float32->float32 and uint16->float32 are)", f.src_dt, f.dst_dt);
}

}  // namespace ipa

#ifdef IPA_FUSED_K
#define IPA_CAT2(a, b) a##b
#define IPA_CAT(a, b) IPA_CAT2(a, b)
int IPA_CAT(ipa_fused_launch_k, IPA_FUSED_K)(ipa_ctx* ctx, const ipa::FusedCall& f) {
  return ipa::fused_launch_k<IPA_FUSED_K>(ctx, f);
}
#endif
