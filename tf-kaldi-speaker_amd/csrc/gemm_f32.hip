// fp32 MFMA GEMM for gfx950 (MI355X): temporal convolution / dense layers of the TDNN as an
// "overlapping-row" GEMM with a fused bias + batch-norm + activation epilogue.
//
//   Y[rowmap[m], n] = act((sum_k A[m,k] * Wt[n,k]) * scale[n] + shift[n]),  A[m,k] = X[m*ldx + k]
//
// replaces tf.layers.conv2d (1,w) / tf.layers.dense + tf.layers.batch_normalization + relu
// (model/tdnn.py:42-130,137-179) with one launch per layer.
//
// Design (CDNA4):
//  * v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate (bitwise an fmaf chain), 64
//    FLOP/clk/SIMD -> 157 TFLOP/s chip peak.  A wave owns a 64x64 output tile = 2x2 MFMA tiles
//    (64 accumulator VGPRs); a 256-thread workgroup owns 128x128; two workgroups per CU.
//  * The K index of an MFMA is only a summation label, so both operands use the same permuted k
//    order: lane (r, h) reads FOUR consecutive k (one ds_read_b128) and feeds them to four
//    successive MFMAs (k = 8q + 4h + j).  One b128 LDS read per operand per 4 MFMAs.
//  * LDS tiles are [row][32 + 4 pad] floats: the 144-byte row stride makes every ds_read_b128
//    lane group hit 16 distinct 16-byte slots (conflict free) and keeps ds_write_b128 aligned.
//  * Register-staged double buffering: the global loads of tile t+1 are issued before the
//    MFMAs of tile t and written to the other LDS buffer after them; one barrier per K tile.
//  * Weights are pre-packed [Npad][Kpad] (k contiguous, zero padded), so the B tile is loaded
//    exactly like the A tile and needs no bounds checks.
//  * XCD-aware block order: consecutive ids on one XCD (id % 8) walk the N tiles of the same M
//    tile, so the A panel is fetched once per XCD L2.
#include <mutex>

#include "xv_epilogue.h"

namespace xv {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDT = BK + 4;               // padded LDS row (floats)
constexpr int TILE_F = BM * LDT;          // floats per operand tile

// FORM: 0 = any shape / alignment (scalar loads), 1 = aligned 1-D rows, 2 = aligned grid form (ResNet taps).  The grid
// addressing is a template case of its own: as a run-time branch in the 1-D loader it cost the TDNN layers 12 %.
template <int FORM>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs p, int nMt, int nNt, int kper) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][BM][LDT]
  float* Bs = smem + 2 * TILE_F;    // [2][BN][LDT]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;

  const int ntiles = nMt * nNt;
  const int slice = blockIdx.x / ntiles;                       // split-K slice (0 when off)
  const int tile = xcd_remap(blockIdx.x - slice * ntiles, ntiles);
  const int mt = tile / nNt, nt = tile - mt * nNt;
  const int m0 = mt * BM, n0 = nt * BN;

  // staging map: thread -> (row lr + 32*i, float4 column c4)
  const int c4 = tid & 7, lr = tid >> 3;
  const int kt0 = slice * kper;
  const int nk = min(p.Kpad / BK, kt0 + kper);
  if (p.ksplit > 1) {                                           // raw partial sums of this slice
    p.Y = p.partial + (int64_t)slice * p.M * p.Npad;
    p.ldy = p.Npad;
    p.N = p.Npad;
    p.Ysb = nullptr;
    p.rowmap = nullptr;
    p.raw = 1;
    p.act = ACT_NONE;
    p.R = nullptr;                       // the shortcut is added once, by splitk_reduce_kernel
  }

  f32x4 ra[4], rb[4];
  auto load_tiles = [&](int kt) {
    const int k = kt * BK + c4 * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = lr + 32 * i;
      rb[i] = *reinterpret_cast<const f32x4*>(p.Wt + (int64_t)(n0 + row) * p.Kpad + k);
      if (FORM != 0) {
        if (k >= p.K) {
          ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else if (FORM == 2) {        // grid form: A[m, tap*ktap + kk] = X[a_off + m*a_pitch + tap*tap_stride + kk]
          const int tap = k / p.ktap;
          ra[i] = *reinterpret_cast<const f32x4*>(p.X + p.a_off + (int64_t)(m0 + row) * p.a_pitch +
                                                  (int64_t)tap * p.tap_stride + (k - tap * p.ktap));
        } else {
          ra[i] = *reinterpret_cast<const f32x4*>(p.X + (int64_t)(m0 + row) * p.ldx + k);
        }
      } else {
        const int m = m0 + row;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < p.M) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int kk = k + e;
            if (kk < p.K) {
              const int dr = kk / p.cin;
              v[e] = p.X[(int64_t)(m + dr) * p.ldx + (kk - dr * p.cin)];
            }
          }
        }
        ra[i] = v;
      }
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = lr + 32 * i;
      *reinterpret_cast<f32x4*>(As + buf * TILE_F + row * LDT + c4 * 4) = ra[i];
      *reinterpret_cast<f32x4*>(Bs + buf * TILE_F + row * LDT + c4 * 4) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  load_tiles(kt0);
  store_tiles(0);
  __syncthreads();

  const float* a_base = As + (wm * 64 + r32) * LDT + 4 * h;
  const float* b_base = Bs + (wn * 64 + r32) * LDT + 4 * h;

  for (int kt = kt0; kt < nk; ++kt) {
    const int cur = (kt - kt0) & 1;
    if (kt + 1 < nk) load_tiles(kt + 1);
    const float* ap = a_base + cur * TILE_F;
    const float* bp = b_base + cur * TILE_F;
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
      f32x4 a0 = *reinterpret_cast<const f32x4*>(ap + q * 8);
      f32x4 a1 = *reinterpret_cast<const f32x4*>(ap + 32 * LDT + q * 8);
      f32x4 b0 = *reinterpret_cast<const f32x4*>(bp + q * 8);
      f32x4 b1 = *reinterpret_cast<const f32x4*>(bp + 32 * LDT + q * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // weights = MFMA A operand, activations = B operand: acc[ni][mi] is D[n][m] (xv_epilogue.h)
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[j], a0[j], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[j], a1[j], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[j], a0[j], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[j], a1[j], acc[1][1], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // the final barrier of the K loop has retired every LDS read: reuse the tiles as store scratch
  store_wave_tile(p, acc, m0 + wm * 64, n0 + wn * 64, lane, wave, reinterpret_cast<char*>(smem));
}

// out[m][n] = act((sum_s partial[s][m][n]) * scale[n] + shift[n]); slices summed in index order.
__global__ void splitk_reduce_kernel(GemmArgs p) {
  const int64_t total = (int64_t)p.M * p.N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / p.N), n = (int)(i - (int64_t)m * p.N);
    float acc = 0.f;
    for (int s = 0; s < p.ksplit; ++s) acc += p.partial[((int64_t)s * p.M + m) * p.Npad + n];
    bool zero;
    const int orow = out_row(p, m, zero);
    if (orow < 0) continue;
    float v = fmaf(acc, p.scale[n], p.shift[n]);
    if (p.R) v += p.R[(int64_t)orow * p.ldr + n];          // residual shortcut before the activation
    p.Y[(int64_t)orow * p.ldy + n] = zero ? 0.f : apply_act(v, p.act, p.alpha ? p.alpha[n] : 0.f);
  }
}

int gemm_f32_ksplit(int M, int Kpad, int Npad) {
  if (M > 512) return 1;                       // frame-level layers fill the chip on their own
  const int tiles = ((M + BM - 1) / BM) * (Npad / BN), nk = Kpad / BK;
  int s = 512 / (tiles > 0 ? tiles : 1);
  if (s > nk / 4) s = nk / 4;                  // at least four K tiles per slice
  return s < 2 ? 1 : s;
}

hipError_t launch_gemm_f32(const GemmArgs& a, bool aligned, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  const int nMt = (a.M + BM - 1) / BM, nNt = a.Npad / BN;
  const int ks = (a.ksplit > 1 && a.partial && !a.Ysb && !a.pool_part) ? a.ksplit : 1;
  const int nk_all = a.Kpad / BK;
  const int kper = (nk_all + ks - 1) / ks;
  const size_t smem = (size_t)4 * TILE_F * sizeof(float);
  static std::mutex init_mu;    // per-device attributes; any thread may make the first launch on a device
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  std::lock_guard<std::mutex> init_lock(init_mu);
  if (!attr_set[dev & 63]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<0>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    attr_set[dev & 63] = true;
  }
  GemmArgs b = a;
  b.ksplit = ks;
  dim3 grid(nMt * nNt * ks), block(256);
  if (aligned && a.a_pitch)
    hipLaunchKernelGGL(gemm_f32_kernel<2>, grid, block, smem, s, b, nMt, nNt, kper);
  else if (aligned)
    hipLaunchKernelGGL(gemm_f32_kernel<1>, grid, block, smem, s, b, nMt, nNt, kper);
  else
    hipLaunchKernelGGL(gemm_f32_kernel<0>, grid, block, smem, s, b, nMt, nNt, kper);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || ks == 1) return e;
  const int64_t total = (int64_t)a.M * a.N;
  const int blocks = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, b);
  return hipGetLastError();
}

}  // namespace xv
