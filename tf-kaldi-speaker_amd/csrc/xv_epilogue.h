// Shared GEMM epilogue for the 32x32 MFMA accumulator layout (gfx950), and the "split-blocked"
// (SB) activation format of the bf16x3 path.
//
// SB format: a row of C channels is stored as ceil(C/32) blocks of 128 bytes; block j holds
// channels 32j..32j+31 as [32 x bf16 hi | 32 x bf16 lo] with hi = rn_bf16(x), lo = rn_bf16(x - hi)
// (hi + lo carries 16 significant bits).  Same bytes per element as fp32; one (row, block) is
// one full 128-byte line, and 16-byte chunk q of a block is plane q>>2, k-step (q>>1)&1,
// lane-half q&1 of the v_mfma_f32_32x32x16_bf16 A/B fragments.
#pragma once
#include "xv_kernels.h"

namespace xv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float apply_act(float v, int act, float alpha) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.0f);
    case ACT_LRELU: return fmaxf(v, kLreluAlpha * v);                       // tf.nn.leaky_relu
    case ACT_PRELU: return fmaxf(v, 0.0f) + alpha * (v - fabsf(v)) * 0.5f;  // model/common.py:40-42
    case ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// round-to-nearest-even fp32 -> bf16 bits (finite inputs; NaN propagates as a quiet NaN)
__device__ __forceinline__ uint32_t bf16_rn_bits(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// (hi | lo << 16) split of one fp32 value
__device__ __forceinline__ uint32_t split_pack(float v) {
  const uint32_t hi = bf16_rn_bits(v);
  const float r = v - __uint_as_float(hi << 16);
  return hi | (bf16_rn_bits(r) << 16);
}

// XCD-aware bijective remap of a 1-D grid (ids congruent mod 8 share an XCD and its L2).
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7, i = id >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// Store one 32x32 accumulator tile.  C/D layout: col = lane & 31, row = (e & 3) + 8 * (e >> 2) +
// 4 * (lane >> 5).  `mbase`/`nbase` are the tile's first row / column (nbase % 32 == 0).
// Writes fp32 Y and/or the SB planes; for SB also zero-fills the padding columns [N, ldsb).
__device__ __forceinline__ void store_tile_32x32(const GemmArgs& p, const f32x16& acc, int mbase, int nbase,
                                                 int lane) {
  const int r32 = lane & 31, h = lane >> 5;
  const int n = nbase + r32;
  const bool nok = n < p.N;
  const float sc = nok ? p.scale[n] : 0.f;
  const float sh = nok ? p.shift[n] : 0.f;
  const float al = (nok && p.alpha) ? p.alpha[n] : 0.f;
  const bool sb_col = p.Ysb && n < p.ldsb;       // column exists in the SB row (value or zero padding)
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int m = mbase + (e & 3) + 8 * (e >> 2) + 4 * h;
    int orow = -1;
    if (m < p.M) orow = p.rowmap ? p.rowmap[m] : m;
    const float v = nok ? apply_act(fmaf(acc[e], sc, sh), p.act, al) : 0.f;
    if (p.Y && nok && orow >= 0) p.Y[(int64_t)orow * p.ldy + n] = v;
    if (p.Ysb) {
      // pair adjacent columns so every lane stores one dword: even lane -> hi plane, odd -> lo plane
      const uint32_t mine = split_pack(v);
      const uint32_t other = __shfl_xor(mine, 1, 64);
      const uint32_t word = (lane & 1) ? ((other >> 16) | (mine & 0xffff0000u)) : ((mine & 0xffffu) | (other << 16));
      if (sb_col && orow >= 0) {
        char* row = reinterpret_cast<char*>(p.Ysb) + (int64_t)orow * p.ldsb * 4;
        const int blk = n >> 5, c = (n & 31) & ~1;
        *reinterpret_cast<uint32_t*>(row + blk * 128 + (lane & 1) * 64 + c * 2) = word;
      }
    }
  }
}

}  // namespace xv
