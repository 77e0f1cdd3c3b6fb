// Shared pieces of the two-unit split (XV_PREC_F16F6): block-scaled fp6 quantisation helpers and the epilogue that writes a layer's
// output in the activation block format of gemm_f16f6.hip (f16 hi | fp6 codes of hi and lo | their E8M0 scales, 128 bytes per
// (row, 32 channels)).  Used by gemm_f16f6.hip and by the F6-output instantiation of the multi-tap f16 kernel (gemm_bf16x3.hip).
#pragma once
#include "xv_epilogue.h"

namespace xv {

// e2m3 code (6 bits) of x * inv, |x * inv| <= 7.5: sign | 2-bit exponent (bias 1) | 3-bit mantissa; exponent field 0 = subnormal
// (steps of 1/8).  Values below 1 go through 1 + v so that one integer path rounds both ranges (round half up; carries walk into
// the exponent field by themselves).
__device__ __forceinline__ uint32_t e2m3_code(float x, float inv) {
  float v = fminf(fabsf(x) * inv, 7.5f);
  const bool sub = v < 1.f;
  const float t = sub ? v + 1.f : v;
  uint32_t c = ((__float_as_uint(t) + 0x80000u) >> 20) - (126u << 3) - (sub ? 8u : 0u);
  c = c > 31u ? 31u : c;
  return c | (x < 0.f ? 32u : 0u);
}
// scale of a block whose largest magnitude is amax: smallest e with amax * 2^-e <= 7.5; returns 2^-e, E8M0 byte in `byte`
__device__ __forceinline__ float e8m0_of(float amax, uint32_t& byte) {
  if (!(amax > 0.f)) { byte = 0; return 1.f; }
  const float r = amax * (1.f / 7.5f);
  int e = (int)((__float_as_uint(r) + 0x7FFFFFu) >> 23) - 127;          // ceil(log2 r)
  e = e < -126 ? -126 : (e > 126 ? 126 : e);
  byte = (uint32_t)(127 + e);
  return __uint_as_float((uint32_t)(127 - e) << 23);
}


// One row per lane: the lane's 32 staged fp32 values (row `rp` of the wave's scratch, chunks swizzled by `sw`) -> the 128-byte block of
// the activation format, written back over the row: hi = rn_f16(x), lo = x - hi, the two block maxima, the fp6 codes packed by the
// hardware converters (v_cvt_scalef32_pk32_fp6_f16 for hi: codes in element order, x / scale, RNE, saturating;
// v_cvt_scalef32_2xpk16_fp6_f32 for lo: code 2i from the first operand, 2i + 1 from the second -- tools/proto/cvt_fp6_probe.hip).
typedef _Float16 v32h_t __attribute__((ext_vector_type(32)));
typedef float v16f_t __attribute__((ext_vector_type(16)));
typedef unsigned v6u_t __attribute__((ext_vector_type(6)));

__device__ __forceinline__ void f6_block_of_row(char* rp, int sw, bool check, uint32_t& bad) {
  float x[32];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(rp + ((q ^ sw) << 4));
    x[4 * q] = v[0]; x[4 * q + 1] = v[1]; x[4 * q + 2] = v[2]; x[4 * q + 3] = v[3];
  }
  v32h_t hv;
  v16f_t la, lb;
  float mh = 0.f, ml = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const _Float16 h0 = (_Float16)x[2 * i], h1 = (_Float16)x[2 * i + 1];
    hv[2 * i] = h0; hv[2 * i + 1] = h1;
    const float f0 = (float)h0, f1 = (float)h1;
    la[i] = x[2 * i] - f0; lb[i] = x[2 * i + 1] - f1;
    mh = fmaxf(mh, fmaxf(fabsf(f0), fabsf(f1)));
    ml = fmaxf(ml, fmaxf(fabsf(la[i]), fabsf(lb[i])));
  }
  uint32_t bh, bl;
  (void)e8m0_of(mh, bh);
  (void)e8m0_of(ml, bl);
  const float sh_f = bh ? __uint_as_float(bh << 23) : 1.f, sl_f = bl ? __uint_as_float(bl << 23) : 1.f;
  const v6u_t ch = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hv, sh_f);
  const v6u_t cl = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(la, lb, sl_f);
  uint32_t hw[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const f16x2_t pr = {hv[2 * i], hv[2 * i + 1]};
    hw[i] = __builtin_bit_cast(uint32_t, pr);
  }
  if (check) {                                         // rows past M hold whatever the slack behind the input held
#pragma unroll
    for (int i = 0; i < 16; ++i) ovf_bits(bad, hw[i]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<uint4*>(rp + ((q ^ sw) << 4)) = uint4{hw[4 * q], hw[4 * q + 1], hw[4 * q + 2], hw[4 * q + 3]};
  *reinterpret_cast<uint4*>(rp + ((4 ^ sw) << 4)) = uint4{ch[0], ch[1], ch[2], ch[3]};
  *reinterpret_cast<uint4*>(rp + ((5 ^ sw) << 4)) = uint4{cl[0], cl[1], cl[2], cl[3]};
  const uint32_t sc2 = bh | (bl << 8);                                            // chunk 6 / 7: code tail | scale dword | pad
  *reinterpret_cast<uint4*>(rp + ((6 ^ sw) << 4)) = uint4{ch[4], ch[5], sc2, 0u};
  *reinterpret_cast<uint4*>(rp + ((7 ^ sw) << 4)) = uint4{cl[4], cl[5], sc2, 0u};
}

// Epilogue that writes the output in the activation block format above (the consumer is another two-unit layer): BN scale / shift +
// activation as usual, the 64 x 32 fp32 values of a pass staged row-major in the wave's LDS scratch, then ONE ROW PER LANE
// (the conversion of f6_block_of_row, written out here), and the usual row-contiguous 16-byte stores (row map applied) follow.
__device__ __forceinline__ void store_wave_tile_n32_f6(const GemmArgs& p, const f32x4 (&acc)[8][2], int mbase, int nbase, int lane,
                                                       int wave, char* lds) {
  constexpr int ROWS = 64, NPASS = 2, FPP = 4;
  const int c16 = lane & 15, g4 = lane >> 4;
  char* scratch = lds + wave * (ROWS * 128);
  const int rrow = lane >> 3, rchunk = lane & 7;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  uint32_t bad = 0;
  f32x4 sc[2], sh[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int n4 = nbase + 16 * ct + 4 * g4;
    const bool ok = n4 < p.N;
    sc[ct] = (ok ? *reinterpret_cast<const f32x4*>(p.scale + n4) : z) * p.sb_mul;      // block-format output only: the layer's power-of-two
    sh[ct] = (ok ? *reinterpret_cast<const f32x4*>(p.shift + n4) : z) * p.sb_mul;      // activation scale is folded in (GemmArgs::sb_mul)
  }
  const bool blk_ok = nbase < p.ldsb;
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    int rmap[ROWS / 8];                  // row map entries of the rows this lane stores in this pass, loaded before its stores
#pragma unroll
    for (int it = 0; it < ROWS / 8; ++it) {
      const int m = mbase + ps * ROWS + it * 8 + rrow;
      rmap[it] = m < p.M ? (p.rowmap ? p.rowmap[m] : m) : -1;
    }
#pragma unroll
    for (int fl = 0; fl < FPP; ++fl)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        f32x4 al = z;
        if (p.act == ACT_PRELU && p.alpha && nbase + 16 * ct + 4 * g4 < p.N) al = *reinterpret_cast<const f32x4*>(p.alpha + nbase + 16 * ct + 4 * g4);
        const f32x4& t = acc[ps * FPP + fl][ct];
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = apply_act(fmaf(t[i], sc[ct][i], sh[ct][i]), p.act, al[i]);
        const int row = fl * 16 + c16;
        *reinterpret_cast<f32x4*>(scratch + row * 128 + (((4 * ct + g4) ^ swz8(row)) << 4)) = v;
      }
    wave_lds_sync();
    {   // one row per lane
      char* rp = scratch + lane * 128;
      const int sw = swz8(lane);
      float x[32];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(rp + ((q ^ sw) << 4));
        x[4 * q] = v[0]; x[4 * q + 1] = v[1]; x[4 * q + 2] = v[2]; x[4 * q + 3] = v[3];
      }
      v32h_t hv;
      v16f_t la, lb;
      float mh = 0.f, ml = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const _Float16 h0 = (_Float16)x[2 * i], h1 = (_Float16)x[2 * i + 1];
        hv[2 * i] = h0; hv[2 * i + 1] = h1;
        const float f0 = (float)h0, f1 = (float)h1;
        la[i] = x[2 * i] - f0; lb[i] = x[2 * i + 1] - f1;
        mh = fmaxf(mh, fmaxf(fabsf(f0), fabsf(f1)));
        ml = fmaxf(ml, fmaxf(fabsf(la[i]), fabsf(lb[i])));
      }
      uint32_t bh, bl;
      (void)e8m0_of(mh, bh);
      (void)e8m0_of(ml, bl);
      const float sh_f = bh ? __uint_as_float(bh << 23) : 1.f, sl_f = bl ? __uint_as_float(bl << 23) : 1.f;
      const v6u_t ch = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hv, sh_f);
      const v6u_t cl = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(la, lb, sl_f);
      uint32_t hw[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const f16x2_t pr = {hv[2 * i], hv[2 * i + 1]};
        hw[i] = __builtin_bit_cast(uint32_t, pr);
      }
      const int mrow = mbase + ps * ROWS + lane;
      if (mrow < p.M && (!p.rowmap || p.rowmap[mrow] != -1)) {   // rows past M, and rows nobody stores, may hold whatever lay behind the input
#pragma unroll
        for (int i = 0; i < 16; ++i) ovf_bits(bad, hw[i]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<uint4*>(rp + ((q ^ sw) << 4)) = uint4{hw[4 * q], hw[4 * q + 1], hw[4 * q + 2], hw[4 * q + 3]};
      *reinterpret_cast<uint4*>(rp + ((4 ^ sw) << 4)) = uint4{ch[0], ch[1], ch[2], ch[3]};
      *reinterpret_cast<uint4*>(rp + ((5 ^ sw) << 4)) = uint4{cl[0], cl[1], cl[2], cl[3]};
      const uint32_t sc2 = bh | (bl << 8);                                            // chunk 6 / 7: code tail | scale dword | pad
      *reinterpret_cast<uint4*>(rp + ((6 ^ sw) << 4)) = uint4{ch[4], ch[5], sc2, 0u};
      *reinterpret_cast<uint4*>(rp + ((7 ^ sw) << 4)) = uint4{cl[4], cl[5], sc2, 0u};
    }
    wave_lds_sync();
#pragma unroll 4
    for (int it = 0; it < ROWS / 8; ++it) {
      const int row = it * 8 + rrow;
      f32x4 v = *reinterpret_cast<const f32x4*>(scratch + row * 128 + ((rchunk ^ swz8(row)) << 4));
      int r = rmap[it];
      const bool zero = r < -1;
      r = zero ? -r - 2 : r;
      if (zero) v = z;
      if (r >= 0 && blk_ok) {
        char* dst = reinterpret_cast<char*>(p.Ysb) + (int64_t)r * p.ldsb * 4 + (nbase >> 5) * 128 + rchunk * 16;
        *reinterpret_cast<f32x4*>(dst) = v;
        // Eight zero rows behind the last row of the value: the reader's zero-weight taps (tap >= its width in a group of four)
        // still multiply rows m + w .. m + 7, and behind the last row that is whatever the workspace held -- an E8M0 scale byte
        // of 255 there is a NaN, and NaN x 0 = NaN.  Written here by the lanes that store the last row (GEMM row M - 1 is the
        // last valid frame of the last utterance) instead of a memset per layer and forward.
        if (mbase + ps * ROWS + row == p.M - 1) {
#pragma unroll
          for (int k = 1; k <= 8; ++k) *reinterpret_cast<f32x4*>(dst + (int64_t)k * p.ldsb * 4) = z;
        }
      }
    }
    wave_lds_sync();
  }
  ovf_report_bits(p.ovf, bad);
}

// The ResNet block outputs (gemm_f6v2_kernel, three-tap form): value = act(scale * acc + shift + R[row]) with the optional fp32
// residual R; wanted in fp32 (GemmArgs::Y: the next block's residual) and / or as the split-blocked row or (GemmArgs::ysb_f6) the
// two-unit block.  The pre-activation values are staged; a row-contiguous sweep adds the residual, activates, stores Y and either
// the split-blocked halves, or writes value * sb_mul back for the row-per-lane conversion and the block stores.  The row map is
// read where it is used (no per-lane copy: the registers around this epilogue are spoken for).
__device__ __forceinline__ void store_wave_tile_n32_res(const GemmArgs& p, const f32x4 (&acc)[8][2], int mbase, int nbase, int lane,
                                                        int wave, char* lds) {
  constexpr int ROWS = 64, NPASS = 2, FPP = 4;
  const int c16 = lane & 15, g4 = lane >> 4;
  char* scratch = lds + wave * (ROWS * 128);
  const int rrow = lane >> 3, rchunk = lane & 7;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  uint32_t bad = 0;
  const int n = nbase + rchunk * 4;
  const bool nok = n < p.N, blk_ok = nbase < p.ldsb;
  auto row_of = [&](int m, bool& zero) -> int {
    int r = m < p.M ? (p.rowmap ? p.rowmap[m] : m) : -1;
    zero = r < -1;
    return zero ? -r - 2 : r;
  };
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int n4 = nbase + 16 * ct + 4 * g4;
      const bool ok = n4 < p.N;
      const f32x4 sc = ok ? *reinterpret_cast<const f32x4*>(p.scale + n4) : z, sh = ok ? *reinterpret_cast<const f32x4*>(p.shift + n4) : z;
#pragma unroll
      for (int fl = 0; fl < FPP; ++fl) {
        const f32x4& t = acc[ps * FPP + fl][ct];
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaf(t[i], sc[i], sh[i]);
        const int row = fl * 16 + c16;
        *reinterpret_cast<f32x4*>(scratch + row * 128 + (((4 * ct + g4) ^ swz8(row)) << 4)) = v;
      }
    }
    wave_lds_sync();
    {
      f32x4 al = z;
      if (p.act == ACT_PRELU && p.alpha && nok) al = *reinterpret_cast<const f32x4*>(p.alpha + n);
#pragma unroll 2
      for (int it = 0; it < ROWS / 8; ++it) {
        const int row = it * 8 + rrow;
        f32x4* slot = reinterpret_cast<f32x4*>(scratch + row * 128 + ((rchunk ^ swz8(row)) << 4));
        f32x4 v = *slot;
        bool zero;
        const int r = row_of(mbase + ps * ROWS + row, zero);
        if (r >= 0 && nok && !zero) {
          if (p.R) v += *reinterpret_cast<const f32x4*>(p.R + (int64_t)r * p.ldr + n);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i], p.act, al[i]);
        } else {
          v = z;
        }
        if (p.Y && r >= 0 && nok) *reinterpret_cast<f32x4*>(p.Y + (int64_t)r * p.ldy + n) = v;
        v *= p.sb_mul;
        if (p.ysb_f6) {
          *slot = v;
        } else if (p.Ysb && r >= 0 && n < p.ldsb) {
          uint32_t h01, l01, h23, l23;
          split2t<true>(v[0], v[1], h01, l01);
          split2t<true>(v[2], v[3], h23, l23);
          ovf_bits(bad, h01); ovf_bits(bad, h23);
          char* blk = reinterpret_cast<char*>(p.Ysb) + (int64_t)r * p.ldsb * 4 + (n >> 5) * 128 + (n & 31) * 2;
          *reinterpret_cast<uint2*>(blk) = make_uint2(h01, h23);
          *reinterpret_cast<uint2*>(blk + 64) = make_uint2(l01, l23);
        }
      }
    }
    if (p.ysb_f6) {
      wave_lds_sync();
      f6_block_of_row(scratch + lane * 128, swz8(lane), true, bad);     // (rows nobody stores were staged as zeros by the sweep)
      wave_lds_sync();
#pragma unroll 2
      for (int it = 0; it < ROWS / 8; ++it) {
        const int row = it * 8 + rrow;
        const f32x4 v = *reinterpret_cast<const f32x4*>(scratch + row * 128 + ((rchunk ^ swz8(row)) << 4));
        bool zero;
        const int r = row_of(mbase + ps * ROWS + row, zero);     // (a zero row was staged as zeros: the all-zero block)
        if (r >= 0 && blk_ok)
          *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(p.Ysb) + (int64_t)r * p.ldsb * 4 + (nbase >> 5) * 128 + rchunk * 16) = v;
      }
    }
    wave_lds_sync();
  }
  ovf_report_bits(p.ovf, bad);
}

}  // namespace xv
