// bf16x3 split-precision MFMA GEMM for gfx950 (MI355X): the TDNN's temporal convolutions and
// dense layers at ~fp32 accuracy on the bf16 matrix pipe.
//
//   x = hi + lo (+ O(2^-17 |x|)),  hi = rn_bf16(x), lo = rn_bf16(x - hi)
//   a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (dropped terms ~2^-17 |a b|)
//
// Three v_mfma_f32_32x32x16_bf16 per product tile, fp32 accumulate: 16/3 = 5.3x the rate of the
// exact v_mfma_f32_32x32x2_f32 path at a relative error of ~1e-5 per layer (measured in
// tests/test_gpu_parity.py against the float64 oracle; the bar is 1e-4).
//
// Both operands arrive in the split-blocked format of xv_epilogue.h: per row, 128-byte blocks
// [32 x bf16 hi | 32 x bf16 lo].  One block is one K step of 32 for one row, so
//   * every global access is a full 128-byte line (8 lanes x 16 bytes),
//   * 16-byte chunk q of a block is exactly the MFMA fragment of plane q>>2, k16-step (q>>1)&1,
//     lane-half q&1 -- one ds_read_b128 per fragment, no repacking anywhere,
//   * a temporal convolution stays an "overlapping-row" GEMM: row m of A is the contiguous
//     run of w*cin/32 blocks that starts at frame m.
// LDS rows are padded to 144 bytes: every ds_read_b128 lane group touches 16 distinct 16-byte
// slots (conflict free).  Register-staged double buffering, one barrier per K step; 128x128
// tile per 256-thread workgroup (2x2 waves of 64x64), two workgroups per CU.
#include "xv_epilogue.h"

namespace xv {

namespace {
constexpr int BM = 128, BN = 128;
constexpr int ROWB = 144;                 // padded LDS bytes per (row, K-step)
constexpr int TILE_B = BM * ROWB;         // bytes per operand tile
typedef int i32x4 __attribute__((ext_vector_type(4)));
}  // namespace

__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(GemmArgs p, int nMt, int nNt) {
  extern __shared__ __attribute__((aligned(16))) char smem3[];
  char* As = smem3;                  // [2][BM][ROWB]
  char* Bs = smem3 + 2 * TILE_B;     // [2][BN][ROWB]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;

  const int tile = xcd_remap(blockIdx.x, nMt * nNt);
  const int mt = tile / nNt, nt = tile - mt * nNt;
  const int m0 = mt * BM, n0 = nt * BN;

  // staging map: thread -> (row lr + 32*i, 16-byte chunk c8 of the 128-byte block)
  const int c8 = tid & 7, lr = tid >> 3;
  const int nk = p.Kpad >> 5;
  const char* Ag = reinterpret_cast<const char*>(p.Xsb) + ((int64_t)(m0 + lr) * p.ldsbx) * 4 + c8 * 16;
  const char* Bg = reinterpret_cast<const char*>(p.Wsb) + ((int64_t)(n0 + lr) * p.Kpad) * 4 + c8 * 16;
  const int64_t a_rs = (int64_t)32 * p.ldsbx * 4, b_rs = (int64_t)32 * p.Kpad * 4;   // 32 rows down

  i32x4 ra[4], rb[4];
  auto load_tiles = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *reinterpret_cast<const i32x4*>(Ag + i * a_rs + (int64_t)kt * 128);
      rb[i] = *reinterpret_cast<const i32x4*>(Bg + i * b_rs + (int64_t)kt * 128);
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int off = (lr + 32 * i) * ROWB + c8 * 16;
      *reinterpret_cast<i32x4*>(As + buf * TILE_B + off) = ra[i];
      *reinterpret_cast<i32x4*>(Bs + buf * TILE_B + off) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  // fragment addresses: chunk = plane*4 + kstep*2 + h
  const char* a_base = As + (wm * 64 + r32) * ROWB + h * 16;
  const char* b_base = Bs + (wn * 64 + r32) * ROWB + h * 16;

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tiles(kt + 1);
    const char* ap = a_base + cur * TILE_B;
    const char* bp = b_base + cur * TILE_B;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8*>(ap + i * 32 * ROWB + s * 32);
        al[i] = *reinterpret_cast<const bf16x8*>(ap + i * 32 * ROWB + 64 + s * 32);
        bh[i] = *reinterpret_cast<const bf16x8*>(bp + i * 32 * ROWB + s * 32);
        bl[i] = *reinterpret_cast<const bf16x8*>(bp + i * 32 * ROWB + 64 + s * 32);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
          // small cross terms first, the dominant hi*hi term last
          acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[nj], acc[mi][nj], 0, 0, 0);
          acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[nj], acc[mi][nj], 0, 0, 0);
          acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[nj], acc[mi][nj], 0, 0, 0);
        }
    }
    if (kt + 1 < nk) store_tiles(cur ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
      store_tile_32x32(p, acc[mi][nj], m0 + wm * 64 + mi * 32, n0 + wn * 64 + nj * 32, lane);
}

hipError_t launch_gemm_bf16x3(const GemmArgs& a, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  const int nMt = (a.M + BM - 1) / BM, nNt = a.Npad / BN;
  const size_t smem = (size_t)4 * TILE_B;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_bf16x3_kernel, dim3(nMt * nNt), dim3(256), smem, s, a, nMt, nNt);
  return hipGetLastError();
}

}  // namespace xv
