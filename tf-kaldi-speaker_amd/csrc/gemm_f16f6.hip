// Two-unit split (XV_PREC_F16F6): the activation block format and its conversion pass.  The GEMM kernel is csrc/gemm_f6v2.hip.
//
//   a * w ~ f16(a) * f16(w)                                                   v_mfma_f32_16x16x32_f16
//         + q6(f16(a)) * q6(w - f16(w)) + q6(a - f16(a)) * q6(f16(w))          2 x v_mfma_scale_f32_16x16x128_f8f6f4
// with q6 = OCP fp6 e2m3 under one power-of-two (E8M0) scale per 32 channels.  The cross terms are ~2^-11 of the result, so the
// 3 mantissa bits of e2m3 leave ~1e-5 relative error per layer (bar 1e-4; tests/analysis/f16f8_error_model.py), and the block-scaled
// instruction runs at four times the f16 rate: a product costs 1 + 2 x 1/4 MFMA units instead of the 3 of bf16x3 / f16x3
// (tools/mfma_mix_bench.hip: 1.6x at the same tile).
//
// Activation block of one (row, 32 channels), 128 bytes like the split-blocked format of xv_epilogue.h:
//   chunks 0-3  32 x f16 hi (unchanged: the main fragment read is the one of the f16x3 kernel)
//   chunk 4 / 5 bytes 0-15 of the 32 x 6-bit codes of q6(hi) / q6(lo)
//   chunk 6 / 7 bytes 16-23 of the codes of q6(hi) / q6(lo) | one dword {byte 0: E8M0 scale of q6(hi), byte 1: of q6(lo)} | 4 bytes pad
// so that the operand of one cross sub-phase (codes + scale of one of the two quantised halves) is two 16-byte reads.  One lane's
// operand of the scaled MFMA is such a block: lane l holds row / column l & 15 and K group l >> 4 = 32 consecutive K elements, 32 x 6
// bits little-endian, scale per lane (checked bit-exactly: tools/mfma_scale_probe.hip).  The K = 128 of a scaled MFMA is four K
// groups, each one (channel block, tap) pair of a quad of channel blocks: a group reads slab row frame + tap of its block -- the rows
// the main MFMA of that tap reads (gemm_f6v2.hip).
#include "xv_f6.h"

namespace xv {

// split-blocked f16 rows (hi | lo halves) -> the block format above; one thread per (row, 32-channel block)
__global__ void f6_from_sb_kernel(const char* __restrict__ sb, char* __restrict__ out, int64_t rows, int nblk) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * nblk) return;
  const uint4* src = reinterpret_cast<const uint4*>(sb + i * 128);
  uint4* dst = reinterpret_cast<uint4*>(out + i * 128);
  uint4 h[4], l[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { h[c] = src[c]; l[c] = src[4 + c]; }
  float hf[32], lf[32];
  float mh = 0.f, ml = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint32_t hw[4] = {h[c].x, h[c].y, h[c].z, h[c].w}, lw[4] = {l[c].x, l[c].y, l[c].z, l[c].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const f16x2_t hh = __builtin_bit_cast(f16x2_t, hw[e]), ll = __builtin_bit_cast(f16x2_t, lw[e]);
      hf[8 * c + 2 * e] = (float)hh[0]; hf[8 * c + 2 * e + 1] = (float)hh[1];
      lf[8 * c + 2 * e] = (float)ll[0]; lf[8 * c + 2 * e + 1] = (float)ll[1];
    }
  }
#pragma unroll
  for (int k = 0; k < 32; ++k) { mh = fmaxf(mh, fabsf(hf[k])); ml = fmaxf(ml, fabsf(lf[k])); }
  uint32_t bh, bl;
  const float ih = e8m0_of(mh, bh), il = e8m0_of(ml, bl);
  uint32_t ch[6], cl[6];                   // 32 x 6 bits = 6 dwords each
#pragma unroll
  for (int d = 0; d < 6; ++d) { ch[d] = 0; cl[d] = 0; }
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const uint32_t a = e2m3_code(hf[k], ih), b = e2m3_code(lf[k], il);
    const int bit = 6 * k, w = bit >> 5, sh = bit & 31;
    ch[w] |= a << sh; cl[w] |= b << sh;
    if (sh > 26) { ch[w + 1] |= a >> (32 - sh); cl[w + 1] |= b >> (32 - sh); }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) dst[c] = h[c];
  dst[4] = uint4{ch[0], ch[1], ch[2], ch[3]};
  dst[5] = uint4{cl[0], cl[1], cl[2], cl[3]};
  const uint32_t sc2 = bh | (bl << 8);
  dst[6] = uint4{ch[4], ch[5], sc2, 0u};
  dst[7] = uint4{cl[4], cl[5], sc2, 0u};
}

hipError_t launch_f6_from_sb(const void* sb, void* out, int64_t rows, int nblk, hipStream_t s) {
  const int64_t total = rows * nblk;
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(f6_from_sb_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, static_cast<const char*>(sb),
                     static_cast<char*>(out), rows, nblk);
  return hipGetLastError();
}

}  // namespace xv
