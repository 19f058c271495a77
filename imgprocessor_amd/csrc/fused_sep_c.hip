// fused_sep_c.hip - the remap ALONE on the marching strips: wave_sep_kernel with K = 1 (no filter, no halo;
// fused_sep_impl.hpp, wave_sep.hpp::sep_geom<1>) - the standalone bilinear remap of frame batches (knob strip_remap)
#include "fused_sep_impl.hpp"

void ipa_fused_sep_launch_c(ipa_ctx* ctx, const ipa::FusedCall& f, const ipa::FusedSep& q) {
  ipa::fused_sep_k<1>(ctx, f, q);
}

// integer frames into their own type with cv2's arithmetic (wave_pipe.hpp CV16: 16U float weights / 8U fixed point): maps,
// every strip on the shared-record loop (fused.hip::ipa_strip_remap_int launches it only then)
template <typename T> static void strip_remap_int_launch(ipa_ctx* ctx, const ipa::FusedCall& f) {
  using namespace ipa;
  using Src = SampleRowSrc<T, kLinear, MapCoord>;
  Src s;
  s.coord = f.map;
  s.src = f.src; s.src_frame_bytes = f.src_frame_bytes; s.src_bytes = f.src_bytes;
  s.sh = f.sh; s.sw = f.sw; s.spitch = f.spitch;
  s.border = f.border; s.q5 = 1; s.cubic_a = f.cubic_a; s.lanczos = nullptr;
  s.cval = (float)f.cval; s.ccval = 0.f; s.map_vec = f.map_vec;
  const double one = 1.0;
  launch_sep<Src, 1, T>(ctx, f.p, s, &one, &one, f.n_frames, 0.f);
}
void ipa_fused_sep_launch_c16(ipa_ctx* ctx, const ipa::FusedCall& f) { strip_remap_int_launch<uint16_t>(ctx, f); }
void ipa_fused_sep_launch_c8(ipa_ctx* ctx, const ipa::FusedCall& f) { strip_remap_int_launch<uint8_t>(ctx, f); }
