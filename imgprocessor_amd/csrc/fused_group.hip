// fused_group.hip - batch form of the fused remap -> K x K filter chain: one workgroup per
// strip of 4 frames, footprints computed once per group, bilinear taps from a per-wave LDS
// ring of source rows (group_stencil.hpp).  Covers bilinear float32 / uint16 frames with any of
// the three coordinate sources; everything else stays on the per-frame kernels.
// Reference call chain: camera/LensDistortion.py:323-326 / camera/PerspectiveCorrection.py:401-405
// followed by filters/maskedConvolve.py:24-43.
#include "fused_impl.hpp"
#include "group_stencil.hpp"

namespace ipa {

// waves per SIMD the register budget is cut for (LDS allows 3 workgroups per CU)
#ifndef IPA_GROUP_MINW
#define IPA_GROUP_MINW 3
#endif

template <typename ST, typename Coord, int K>
__global__ void __launch_bounds__(64 * kGW, IPA_GROUP_MINW)
group_stencil_kernel(WaveParams p, GroupSrc<ST, Coord> g, Weights<float, K * K> wts) {
  GroupKernel<ST, Coord, K, false>::body(p, g, wts, nullptr);
}

template <typename ST, typename Coord, int K> struct GroupBigArgs {
  WaveParams p;
  GroupSrc<ST, Coord> g;
  alignas(16) float wrows[K][12];  // kernel row i, taps 0..K-1, zero padded
};
template <typename ST, typename Coord, int K>
__global__ void __launch_bounds__(64 * kGW, IPA_GROUP_MINW)
group_stencil_big_kernel(GroupBigArgs<ST, Coord, K> a) {
  typedef const char __attribute__((address_space(4)))* kernarg_bytes;
  kernarg_bytes base = (kernarg_bytes)__builtin_amdgcn_kernarg_segment_ptr();
  using Args = GroupBigArgs<ST, Coord, K>;
  kernarg_f32 wk = (kernarg_f32)(base + offsetof(Args, wrows));
  Weights<float, K * K> unused;
  GroupKernel<ST, Coord, K, true>::body(a.p, a.g, unused, wk);
}

template <typename ST, typename Coord>
static void group_fill(const FusedCall& f, const Coord& c, GroupSrc<ST, Coord>& g, int use_ring) {
  g.coord = c;
  g.src = f.src; g.src_frame_bytes = f.src_frame_bytes; g.src_bytes = f.src_bytes;
  g.sh = f.sh; g.sw = f.sw; g.spitch = f.spitch;
  g.border = f.border; g.q5 = f.q5;
  g.cval = (float)f.cval; g.ccval = (float)f.conv_cval;
  g.n_frames = f.n_frames;
  g.use_ring = use_ring;
}

template <int K>
static dim3 group_grid(const ipa_ctx* ctx, WaveParams& p, int n_frames, bool fma_bound) {
  using G = group_geom<K>;
  p.strips_x = (p.dw + G::OW - 1) / G::OW;
  // same rule as the per-frame kernels (tallest strip that still leaves enough waves), counted
  // with this kernel's 128-px strips: pretend the image is twice as wide
  p.strip_h = wave_strip_height(ctx, p.dh, 2 * p.dw, n_frames, K, fma_bound);
  p.strips = (unsigned)p.strips_x * (unsigned)((p.dh + p.strip_h - 1) / p.strip_h);
  p.frames_inner = 0;
  const unsigned groups = ((unsigned)n_frames + kGW - 1) / kGW;
  return dim3(p.strips * groups, 1);
}

template <typename ST, typename Coord, int K>
static void group_launch(ipa_ctx* ctx, const FusedCall& f, const Coord& c, int use_ring) {
  if constexpr (K <= 5) {
    Weights<float, K * K> w;
    for (int i = 0; i < K * K; i++) w.w[i] = (float)f.kernel[i];
    GroupSrc<ST, Coord> g;
    group_fill<ST, Coord>(f, c, g, use_ring);
    WaveParams p = f.p;
    dim3 grid = group_grid<K>(ctx, p, f.n_frames, false);
    hipLaunchKernelGGL((group_stencil_kernel<ST, Coord, K>), grid, dim3(64 * kGW), 0, ctx->stream,
                       p, g, w);
  } else {
    GroupBigArgs<ST, Coord, K> a;
    for (int i = 0; i < K; i++)
      for (int j = 0; j < 12; j++) a.wrows[i][j] = j < K ? (float)f.kernel[i * K + j] : 0.f;
    group_fill<ST, Coord>(f, c, a.g, use_ring);
    a.p = f.p;
    dim3 grid = group_grid<K>(ctx, a.p, f.n_frames, K >= 9);
    hipLaunchKernelGGL((group_stencil_big_kernel<ST, Coord, K>), grid, dim3(64 * kGW), 0,
                       ctx->stream, a);
  }
}

template <typename ST, int K>
static int group_launch_coord(ipa_ctx* ctx, const FusedCall& f, int use_ring) {
  switch (f.coord_kind) {
    case 0: group_launch<ST, MapCoord, K>(ctx, f, f.map, use_ring); return 0;
    case 1: group_launch<ST, UndistortCoord, K>(ctx, f, f.und, use_ring); return 0;
    default: group_launch<ST, HomographyCoord, K>(ctx, f, f.hom, use_ring); return 0;
  }
}

}  // namespace ipa

// returns 1 when the call is not covered (the caller then uses the per-frame kernels)
int ipa_fused_group_launch(ipa_ctx* ctx, const ipa::FusedCall& f, int K, int use_ring) {
  using namespace ipa;
  // exact-coordinate bilinear only (the 1/32-px rule of cv2 stays on the per-frame kernels)
  if (f.dst_dt != IPA_F32 || f.interp_base != IPA_INTER_LINEAR || f.q5) return 1;
  if (f.src_dt == IPA_F32) {
    switch (K) {
      case 5: return group_launch_coord<float, 5>(ctx, f, use_ring);
      default: return 1;
    }
  }
  return 1;
}
