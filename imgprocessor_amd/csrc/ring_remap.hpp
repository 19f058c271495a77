// ring_remap.hpp — the standalone remap (cv2.remap / cv2.warpPerspective) for batches with the
// taps read from LDS: bilinear, bicubic (both Keys parameters, exact or 1/32-px coordinates) and
// Lanczos4 - the reference's DEFAULT perspective interpolation
// (camera/PerspectiveCorrection.py:401-405; uncorrect / distort :377-378, :241-242 use bicubic).
//
// The wide footprints are where the gather kernels of remap_impl.hpp hurt most: a Lanczos4
// sample is 8 rows x 2 dwordx4 gathers through the texture addresser (98 % busy, 1.1 ms per
// 16 x 4K).  Here the planning kernel of ring_stencil.hpp (run with the footprint size of the
// interpolation) marks the CLEAN 128-px strips - every footprint inside the source, a two-row
// step within the ring depth, rows moving forward - and this kernel runs them: one wave per
// (strip, frame), source rows by coalesced loads one step ahead into a wave-private LDS ring
// (bilinear / bicubic 8 rows, Lanczos4 16; the first NT-1 slots are mirrored behind the last so a
// footprint never wraps), a sample = NT x NT LDS reads from one computed address + the fma chain
// of sample().  remap_kernel runs the other tiles behind the pair_clean skip mask.
// Same weights (axis_split), same summation order: results identical to remap_kernel.
#pragma once

#include "ring_stencil.hpp"

namespace ipa {

// v_pk_fma_f32 / v_pk_mul_f32 with one half of a VGPR weight pair applied to both components
template <int HI> __device__ __forceinline__ v2f pk_fma_half(v2f wp, v2f x, v2f c) {
  v2f d;
  if constexpr (HI)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(wp), "v"(x), "v"(c));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(wp), "v"(x), "v"(c));
  return d;
}
template <int HI> __device__ __forceinline__ v2f pk_mul_half(v2f wp, v2f x) {
  v2f d;
  if constexpr (HI)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(wp), "v"(x));
  else
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(wp), "v"(x));
  return d;
}

// the LDS byte address of a pointer into a __shared__ object
__device__ __forceinline__ int lds_address(const void* p) {
  return (int)(unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
// ds_read2_b32 with both dword offsets given: {base[O0], base[O1]}.  The compiler does not see
// a memory access here: lds_wait_all() before the first use of any result.
template <int O0, int O1> __device__ __forceinline__ v2f lds_read2(int byte_addr) {
  v2f d;
  asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(d) : "v"(byte_addr), "n"(O0), "n"(O1));
  return d;
}
// ds_read_b64 of an 8-byte-aligned address + byte offset (16-bit immediate)
template <int OFF> __device__ __forceinline__ v2f lds_read_b64(int byte_addr) {
  v2f d;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(byte_addr), "n"(OFF));
  return d;
}
__device__ __forceinline__ void lds_wait_all() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int INTERP> struct ring_rows { static constexpr int value = 8; };
template <> struct ring_rows<kLanczos4> { static constexpr int value = 16; };

struct RingRemapArgs {
  char* dst;
  long dst_frame_elems, dpitch;
  const char* src;
  long src_frame_bytes;
  unsigned src_bytes;
  int spitch;
  int n_frames;
  int q5;
  float cubic_a;
  const float* lanczos;  // [32][8] table (device)
};

template <int INTERP, typename Coord> struct RingRemapKernel {
  using C = typename Coord::coord_t;
  static constexpr int NT = ntaps<INTERP>::value;
  static constexpr int RR = ring_rows<INTERP>::value;
  static constexpr int kSlots = RR + NT - 1;
  // Lanczos4: rows in interleaved pairs - pair p = {row 2p, row 2p + 1}, column c of both at
  // float 2c - so that ONE aligned ds_read_b64 (256 B/clk, twice ds_read2_b32) fetches two tap
  // rows of a column.  A footprint starting on an odd row spans 5 pairs; the 4 pairs after the
  // last one mirror the first 4.
  // (no mirror slots: the 5 pair addresses of a sample wrap individually - 10.2 KB per wave
  // instead of 15.4, 14 instead of 10 waves per CU)
  static constexpr int kPairs = RR / 2, kPairSlots = RR / 2, kPairFloats = 2 * kRW;
  static constexpr int kRingFloats = INTERP == kLanczos4 ? kPairSlots * kPairFloats : kSlots * kRW;
  // frames per workgroup: the Lanczos4 ring is 14.7 KB per wave
  static constexpr int kWaves = INTERP == kLanczos4 ? 2 : 4;

  struct Shared {
    float ring[kWaves][kRingFloats];
    float lz[INTERP == kLanczos4 ? 256 : 4];
  };

  static __device__ __forceinline__ void body(const RingGeom& gm, const RingRemapArgs& a,
                                              const Coord& coord, const RingPlan& plan) {
    __shared__ __attribute__((aligned(16))) Shared sh;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (INTERP == kLanczos4) {
      for (unsigned e = threadIdx.x; e < 256u; e += 64u * kWaves) sh.lz[e] = a.lanczos[e];
      __syncthreads();
    }
    const unsigned groups = ((unsigned)a.n_frames + kWaves - 1) / kWaves;
    const unsigned b = xcd_swizzle(blockIdx.x, gridDim.x);
    const unsigned grp = b % groups;
    const unsigned sid = b / groups;
    const unsigned frame = grp * kWaves + wave;
    if (frame >= (unsigned)a.n_frames) return;
    const int syi = (int)(sid / (unsigned)gm.strips_x), sxi = (int)sid - syi * gm.strips_x;
    if (!plan.pair_clean[syi * gm.pairs_x + (sxi >> 1)]) return;  // remap_kernel's tiles
    const int4 info = plan.info[sid];
    const int xlo = __builtin_amdgcn_readfirstlane(info.y);
    const unsigned* cwp = plan.cnts + (size_t)sid * kPlanWords;

    const int xs = sxi * kSW;
    const int y0 = syi * gm.strip_h;
    const int T = gm.dh - y0 < gm.strip_h ? gm.dh - y0 : gm.strip_h;  // output rows
    const int nsteps = (T + 1) / 2;
    float* dst = reinterpret_cast<float*>(a.dst) + (long)frame * a.dst_frame_elems +
                 ((long)y0 * a.dpitch + xs);
    SrcView s;
    s.rsrc = make_rsrc(a.src + (long)frame * a.src_frame_bytes, a.src_bytes);
    s.q5 = a.q5;
    s.cubic_a = a.cubic_a;
    s.lanczos = sh.lz;
    float* ringw = sh.ring[wave];
    auto put_row = [&](const PendRow<float>& r, int y) {
      if constexpr (INTERP == kLanczos4) {
        const int ps = (y >> 1) & (kPairs - 1);
        float* row = ringw + ps * kPairFloats + (y & 1);
        r.write_every_other(row, lane);
      } else {
        const int slot = y & (RR - 1);
        r.write(ringw + slot * kRW, lane);
        if (slot < NT - 1) r.write(ringw + (slot + RR) * kRW, lane);  // mirror: no footprint wraps
      }
    };

    int hres = __builtin_amdgcn_readfirstlane(info.z);  // rows [.., hres) are in the ring
    {
      const int n0 = __builtin_amdgcn_readfirstlane(info.w);
#pragma unroll 1
      for (int j = 0; j < n0; j += 2) {
        PendRow<float> p0, p1;
        p0.load(s.rsrc, __mul24(hres + j, a.spitch) + xlo, lane);
        if (j + 1 < n0) p1.load(s.rsrc, __mul24(hres + j + 1, a.spitch) + xlo, lane);
        put_row(p0, hres + j);
        if (j + 1 < n0) put_row(p1, hres + j + 1);
      }
      hres += n0;
    }
    C cx[4], cy[4];
    auto step_coords = [&](int st) {
      const int sc = st < nsteps ? st : nsteps - 1;
      const int v0 = y0 + 2 * sc;
      const int v1 = 2 * sc + 1 < T ? v0 + 1 : v0;
      ring_coords<Coord>(coord, xs, v0, v1, cx, cy);
    };
    step_coords(0);
    // 4 bits per step: a 32-row strip has 16 steps = 2 words (scalar registers)
    const unsigned cw0 = __builtin_amdgcn_readfirstlane(cwp[0]);
    const unsigned cw1 = __builtin_amdgcn_readfirstlane(cwp[1]);
    PendRow<float> pend[kRingMaxNew];
    int cnt = 0;

#pragma unroll 1
    for (int st = 0; st < nsteps; st++) {
      // 1. the rows requested during the previous step go into the ring
      cnt = __builtin_amdgcn_readfirstlane(cnt);
      hres = __builtin_amdgcn_readfirstlane(hres);
#pragma unroll
      for (int j = 0; j < kRingMaxNew; j++)
        if (j < cnt) put_row(pend[j], hres + j);
      hres += cnt;
      // 2. footprints and weights of this step (the arithmetic of sample()); Lanczos keeps the
      //    table rows' indices and fetches the 16 weights of a sample next to its taps
      constexpr bool kLz = INTERP == kLanczos4;
      int ad[4];
      float wx[4][kLz ? 1 : NT], wy[4][kLz ? 1 : NT];
      int kx[4], ky[4];
      bool odd[4];  // Lanczos4: the footprint starts on the second row of its first pair
#pragma unroll
      for (int k = 0; k < 4; k++) {
        int ix0, iy0;
        if constexpr (kLz) {
          const int qx = (int)ipa_rint(cx[k] * (C)32), qy = (int)ipa_rint(cy[k] * (C)32);
          ix0 = (qx >> 5) - 3;
          iy0 = (qy >> 5) - 3;
          kx[k] = (qx & 31) << 5;  // byte offset of the table row
          ky[k] = (qy & 31) << 5;
          odd[k] = (iy0 & 1) != 0;
          ad[k] = (iy0 >> 1) | ((ix0 - xlo) << 16);  // first pair | column (both < 2^15)
        } else {
          axis_split<INTERP, float, C>(s, cx[k], ix0, wx[k]);
          axis_split<INTERP, float, C>(s, cy[k], iy0, wy[k]);
          ad[k] = (__mul24(iy0 & (RR - 1), kRW) + (ix0 - xlo)) << 2;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; k++) asm volatile("" : "+v"(ad[k]) : : "memory");
      // 3. requests for the next step
      const int sn = st + 1;
      cnt = sn < nsteps ? (int)((((sn >> 3) ? cw1 : cw0) >> (4 * (sn & 7))) & 15u) : 0;
      cnt = __builtin_amdgcn_readfirstlane(cnt);
#pragma unroll
      for (int j = 0; j < kRingMaxNew; j++)
        if (j < cnt) pend[j].load(s.rsrc, __mul24(hres + j, a.spitch) + xlo, lane);
      step_coords(sn);
      __builtin_amdgcn_sched_barrier(0);

      // 4. taps from the ring, separable weighted sum in the order of sample().  Two tap rows at
      //    a time: ds_read2_b32 fetches {row r, row r + 1} of a column into a register pair and
      //    v_pk_fma_f32 applies the column weight (one half of a weight pair, op_sel) to both -
      //    per component the products and sums of sample(), in its order.  The reads are written
      //    out (the compiler pairs neighbouring columns, which costs two moves per product);
      //    kBatch samples' reads are in flight, then one wait.
      __builtin_amdgcn_wave_barrier();
      const int rbase = lds_address(ringw);
      float out[4];
      if constexpr (kLz) {
        // (the reads of sample k + 1 issued between the fmas of sample k, two register sets:
        // 198 VGPRs, 642 -> 708 us - one sample at a time it stays)
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const char* lzb = reinterpret_cast<const char*>(sh.lz);
          const float4* rx = reinterpret_cast<const float4*>(lzb + kx[k]);
          const float4* ry = reinterpret_cast<const float4*>(lzb + ky[k]);
          const float4 a0 = rx[0], a1 = rx[1], b0 = ry[0], b1 = ry[1];
          const v2f wp[4] = {v2f{a0.x, a0.y}, v2f{a0.z, a0.w}, v2f{a1.x, a1.y}, v2f{a1.z, a1.w}};
          const float uy[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
          // 5 pairs x 8 columns: rows 2 pb .. 2 pb + 9, of which the sample uses 8 from row
          // `odd` on (the other two are never selected)
          v2f t[5][8];
          const int pb = ad[k] & 0xffff, cb = rbase + ((ad[k] >> 16) << 3);
          static_for<0, 5>([&](auto pp_) {
            constexpr int pp = decltype(pp_)::value;
            const int ra = cb + __mul24((pb + pp) & (kPairs - 1), kPairFloats * 4);
            static_for<0, 8>([&](auto cc) {
              constexpr int c = decltype(cc)::value;
              t[pp][c] = lds_read_b64<c * 8>(ra);
            });
          });
          lds_wait_all();
#pragma unroll
          for (int pp = 0; pp < 5; pp++)
#pragma unroll
            for (int c = 0; c < 8; c++) asm volatile("" : "+v"(t[pp][c]));
          float rsum[10];
#pragma unroll
          for (int pp = 0; pp < 5; pp++) {
            v2f rs = pk_mul_half<0>(wp[0], t[pp][0]);
#pragma unroll
            for (int c = 1; c < 8; c++) {
              if (c & 1) rs = pk_fma_half<1>(wp[c >> 1], t[pp][c], rs);
              else rs = pk_fma_half<0>(wp[c >> 1], t[pp][c], rs);
            }
            rsum[2 * pp] = rs.x;
            rsum[2 * pp + 1] = rs.y;
          }
          float o = uy[0] * (odd[k] ? rsum[1] : rsum[0]);
#pragma unroll
          for (int r = 1; r < 8; r++) o = ipa_fma(uy[r], odd[k] ? rsum[r + 1] : rsum[r], o);
          asm volatile("" : "+v"(o));
          out[k] = o;
          __builtin_amdgcn_sched_barrier(0);  // one sample's taps in flight
        }
      } else {
      constexpr int kBatch = 4;
#pragma unroll
      for (int k0 = 0; k0 < 4; k0 += kBatch) {
        float ux[kBatch][NT], uy[kBatch][NT];
        v2f t[kBatch][NT / 2][NT];
#pragma unroll
        for (int q = 0; q < kBatch; q++) {
          const int k = k0 + q;
          if constexpr (kLz) {
            const char* lzb = reinterpret_cast<const char*>(sh.lz);
            const float4* rx = reinterpret_cast<const float4*>(lzb + kx[k]);
            const float4* ry = reinterpret_cast<const float4*>(lzb + ky[k]);
            const float4 a0 = rx[0], a1 = rx[1], b0 = ry[0], b1 = ry[1];
            ux[q][0] = a0.x; ux[q][1] = a0.y; ux[q][2 % NT] = a0.z; ux[q][3 % NT] = a0.w;
            ux[q][4 % NT] = a1.x; ux[q][5 % NT] = a1.y; ux[q][6 % NT] = a1.z; ux[q][7 % NT] = a1.w;
            uy[q][0] = b0.x; uy[q][1] = b0.y; uy[q][2 % NT] = b0.z; uy[q][3 % NT] = b0.w;
            uy[q][4 % NT] = b1.x; uy[q][5 % NT] = b1.y; uy[q][6 % NT] = b1.z; uy[q][7 % NT] = b1.w;
          } else {
#pragma unroll
            for (int c = 0; c < NT; c++) { ux[q][c] = wx[k][c]; uy[q][c] = wy[k][c]; }
          }
#pragma unroll
          for (int rp = 0; rp < NT / 2; rp++) {
            const int ra = rbase + ad[k] + rp * (2 * kRW * 4);
            static_for<0, NT>([&](auto cc) {
              constexpr int c = decltype(cc)::value;
              t[q][rp][c] = lds_read2<c, c + kRW>(ra);
            });
          }
        }
        lds_wait_all();
#pragma unroll
        for (int q = 0; q < kBatch; q++)
#pragma unroll
          for (int rp = 0; rp < NT / 2; rp++)
#pragma unroll
            for (int c = 0; c < NT; c++) asm volatile("" : "+v"(t[q][rp][c]));
#pragma unroll
        for (int q = 0; q < kBatch; q++) {
          v2f wp[NT / 2];
#pragma unroll
          for (int j = 0; j < NT / 2; j++) wp[j] = v2f{ux[q][2 * j], ux[q][2 * j + 1]};
          float o = 0.f;
#pragma unroll
          for (int rp = 0; rp < NT / 2; rp++) {
            v2f rs = pk_mul_half<0>(wp[0], t[q][rp][0]);
#pragma unroll
            for (int c = 1; c < NT; c++) {
              if (c & 1) rs = pk_fma_half<1>(wp[c >> 1], t[q][rp][c], rs);
              else rs = pk_fma_half<0>(wp[c >> 1], t[q][rp][c], rs);
            }
            o = rp == 0 ? uy[q][0] * rs.x : ipa_fma(uy[q][2 * rp], rs.x, o);
            o = ipa_fma(uy[q][2 * rp + 1], rs.y, o);
          }
          out[k0 + q] = o;
        }
        if constexpr (kLz) __builtin_amdgcn_sched_barrier(0);  // one sample's taps in flight
      }
      }
      __builtin_amdgcn_wave_barrier();
      // 5. stores: sample k = row (k >> 1), column lane + 64 (k & 1)
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int row = 2 * st + (k >> 1);
        if (row < T) dst[(long)row * a.dpitch + 64u * (k & 1) + lane] = out[k];
      }
    }
  }
};

template <int INTERP>
struct ring_remap_block {
  static constexpr int value = 64 * (INTERP == kLanczos4 ? 2 : 4);
};

template <int INTERP, typename Coord>
__global__ void __launch_bounds__(ring_remap_block<INTERP>::value)
ring_remap_kernel(RingGeom gm, RingRemapArgs a, Coord coord, RingPlan plan) {
  RingRemapKernel<INTERP, Coord>::body(gm, a, coord, plan);
}

}  // namespace ipa
