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
#include <cstdlib>

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
          // weights = MFMA A operand, activations = B operand: acc[nj][mi] is D[n][m] (xv_epilogue.h);
          // small cross terms first, the dominant hi*hi term last
          acc[nj][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[nj], al[mi], acc[nj][mi], 0, 0, 0);
          acc[nj][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[nj], ah[mi], acc[nj][mi], 0, 0, 0);
          acc[nj][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[nj], ah[mi], acc[nj][mi], 0, 0, 0);
        }
    }
    if (kt + 1 < nk) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // the final barrier of the K loop has retired every LDS read: reuse the tiles as store scratch
  store_wave_tile(p, acc, m0 + wm * 64, n0 + wn * 64, lane, wave, smem3);
}

// ------------------------------------------------------------------------------------------
// 128x128 LDS-DMA kernel (the default): same 2x2 waves of 64x64, two workgroups per CU, but
//  * operands go global -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write: the
//    register-staged kernel above spends ~55% of the LDS port on ds_write_b128 at full MFMA rate);
//  * LDS rows are unpadded 128-byte blocks (the DMA writes 64 lanes x 16 B linearly), made
//    conflict-free by an XOR swizzle applied to the per-lane SOURCE address and to the fragment
//    read address: chunk c of row r lives at r*128 + ((c ^ ((r >> 1) & 7)) << 4);
//  * convolutions reuse the A slab (frames [m0, m0+128+w-1), one 32-channel block) for all w
//    taps by shifting the fragment row, so A is staged once per channel block (traffic / w).
// Schedule per step (one tap of one channel block): issue the DMA of the next step's weight
// tile and this step's share of the next slab, read 16 fragments, 24 MFMAs, __syncthreads()
// (whose vmcnt(0) retires the DMA issued ~800 cycles earlier).
namespace {
constexpr int DROW = 128;                          // unpadded LDS row
constexpr int DA_ROWS = 136;                       // 128 + (w-1 <= 7) halo rows, multiple of 8
constexpr int DA_BYTES = DA_ROWS * DROW;
constexpr int DB_BYTES = BN * DROW;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
}  // namespace

// (A warp-specialised form -- consumer waves + dedicated DMA loader waves -- was measured at 2x SLOWER and
// removed: profiles/README.md.)
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_dma_kernel(GemmArgs p, int nMt, int nNt, int w, int diag) {
  extern __shared__ __attribute__((aligned(16))) char smem3[];
  char* As = smem3;                      // [2][DA_ROWS][128]
  char* Bs = smem3 + 2 * DA_BYTES;       // [2][BN][128]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NL = 4;                            // every wave issues its share of the DMA pieces
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;

  const int tile = xcd_remap(blockIdx.x, nMt * nNt);
  const int mt = tile / nNt, nt = tile - mt * nNt;
  const int m0 = mt * BM, n0 = nt * BN;

  const int ncb = (p.Kpad >> 5) / w;               // channel blocks per frame
  const int nsteps = (diag & 1) ? 0 : ncb * w;     // diag bit0: skip the main loop (timing only)
  const int ngroups = (BM + w - 1 + 7) >> 3;        // 8-row DMA groups per slab (16 or 17)
  const int gps = (ngroups + w - 1) / w;            // groups issued per step
  // per-lane DMA source: row (lane >> 3) of the group, swizzled chunk
  const int lrow = lane >> 3, lpc = lane & 7;
  // A addressing: default 1-D form (row pitch = ldsbx) or the grid form of GemmArgs (pitch / offset /
  // taps with a stride; then w == 1 and K step s is (tap, block) = (s / kbt, s % kbt))
  const int64_t a_row_bytes = (p.a_pitch ? p.a_pitch : p.ldsbx) * 4, b_row_bytes = (int64_t)p.Kpad * 4;
  const int kbt = p.a_pitch ? (p.ktap >> 5) : (p.Kpad >> 5);       // K blocks per tap
  const int64_t tap_bytes = p.a_pitch ? p.tap_stride * 4 : 0;
  const char* Ag = reinterpret_cast<const char*>(p.Xsb) + p.a_off * 4 + (int64_t)(m0 + lrow) * a_row_bytes;
  const char* Bg = reinterpret_cast<const char*>(p.Wsb) + (int64_t)(n0 + lrow) * b_row_bytes;

  // group g covers rows 8g .. 8g+7; this lane's row is 8g + lrow and (row >> 1) & 7 == (4g + (lrow >> 1)) & 7
  // `koff` = byte offset of the K block inside a row: cb*128 in the 1-D form, tap*tap_bytes + blk*128
  // in the grid form (kept incrementally by the caller: no division in the loop)
  auto dma_a = [&](int64_t koff, int buf, int g) {
    const int c = lpc ^ ((4 * g + (lrow >> 1)) & 7);
    __builtin_amdgcn_global_load_lds((gptr_t)(Ag + (int64_t)(8 * g) * a_row_bytes + koff + c * 16),
                                     (lptr_t)(As + buf * DA_BYTES + g * 1024), 16, 0, 0);
  };
  // weight tile: wave `wv` of NL owns 16/NL CONSECUTIVE 1-KB groups, addressed with ONE M0 value plus the
  // instruction's immediate offset (which the hardware adds to both the LDS and the global address, hence
  // the "- Q * 1024" on the source pointer)
  auto dma_b_all = [&](int kb, int buf, int wv) {
    constexpr int GP = 16 / NL;                        // groups per wave
    char* lbase = Bs + buf * DB_BYTES + wv * GP * 1024;
    const char* gbase = Bg + (int64_t)(8 * GP * wv) * b_row_bytes + (int64_t)kb * 128;
#define XV_DMA_B(Q)                                                                                              \
    if (Q < GP) {                                                                                                \
      const int c = lpc ^ ((4 * (GP * wv + Q) + (lrow >> 1)) & 7);                                               \
      __builtin_amdgcn_global_load_lds((gptr_t)(gbase + (int64_t)(8 * Q) * b_row_bytes + c * 16 - Q * 1024),   \
                                       (lptr_t)lbase, 16, Q * 1024, 0);                                          \
    }
    XV_DMA_B(0) XV_DMA_B(1) XV_DMA_B(2) XV_DMA_B(3)
#undef XV_DMA_B
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // prologue: whole slab 0 and the weight tile of step 0
  for (int g = wave; g < ngroups; g += NL) dma_a(0, 0, g);
  int64_t koff_next = (kbt == 1 && p.a_pitch) ? tap_bytes : 128;   // K-block offset of slab cb + 1
  int blk_next = (kbt == 1 && p.a_pitch) ? 0 : 1;                  // its block index inside the tap
  dma_b_all(0, 0, wave);
  __syncthreads();

  // B fragment offsets (row fixed per lane): chunk c = plane*4 + ks*2 + h
  int boff[2];
  int bswz[2];
#pragma unroll
  for (int nj = 0; nj < 2; ++nj) {
    const int rb = wn * 64 + nj * 32 + r32;
    boff[nj] = rb * DROW;
    bswz[nj] = (rb >> 1) & 7;
  }

  int cb = 0, j = 0;
  for (int s = 0; s < nsteps; ++s) {
    int cb_next = cb, j_next = j + 1;
    if (j_next == w) { j_next = 0; cb_next = cb + 1; }
    if (s + 1 < nsteps && !(diag & 4)) {   // diag bit2: no weight DMA in the loop (timing only)
      const int kb = (diag & 32) ? 0 : j_next * ncb + cb_next;   // diag bit5: always the same (L2-hot) source tile
      dma_b_all(kb, (s + 1) & 1, wave);
    }
    if (cb + 1 < ncb && !(diag & 8)) {     // diag bit3: no slab DMA in the loop (timing only)
      const int gend = min((j + 1) * gps, ngroups);
      for (int g = j * gps + wave; g < gend; g += NL) dma_a((diag & 32) ? 0 : koff_next, (cb + 1) & 1, g);
    }

    const char* ab = As + (cb & 1) * DA_BYTES;
    const char* bb = Bs + (s & 1) * DB_BYTES;
    int aoff[2], aswz[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int ra = wm * 64 + mi * 32 + r32 + j;
      aoff[mi] = ra * DROW;
      aswz[mi] = (ra >> 1) & 7;
    }
    bf16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];     // [ks][tile]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ch = ks * 2 + h, cl = 4 + ks * 2 + h;
        ah[ks][i] = *reinterpret_cast<const bf16x8*>(ab + aoff[i] + ((ch ^ aswz[i]) << 4));
        al[ks][i] = *reinterpret_cast<const bf16x8*>(ab + aoff[i] + ((cl ^ aswz[i]) << 4));
        bh[ks][i] = *reinterpret_cast<const bf16x8*>(bb + boff[i] + ((ch ^ bswz[i]) << 4));
        bl[ks][i] = *reinterpret_cast<const bf16x8*>(bb + boff[i] + ((cl ^ bswz[i]) << 4));
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
          acc[nj][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[ks][nj], al[ks][mi], acc[nj][mi], 0, 0, 0);
          acc[nj][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[ks][nj], ah[ks][mi], acc[nj][mi], 0, 0, 0);
          acc[nj][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[ks][nj], ah[ks][mi], acc[nj][mi], 0, 0, 0);
        }
    if (!(diag & 16)) __syncthreads();     // diag bit4: no barrier (timing only, racy)
    if (cb_next != cb) {                 // advance the A K-block offset with the slab index
      if (++blk_next == kbt) {
        blk_next = 0;
        koff_next += tap_bytes - (int64_t)(kbt - 1) * 128;
      } else {
        koff_next += 128;
      }
    }
    cb = cb_next;
    j = j_next;
  }

  if (diag & 2) {       // diag bit1: skip the epilogue stores (keep the accumulators live)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) asm volatile("" ::"v"(acc[mi][nj]));
    return;
  }
  // the final barrier of the K loop has retired every LDS read: reuse the tiles as store scratch
  store_wave_tile(p, acc, m0 + wm * 64, n0 + wn * 64, lane, wave, smem3);
}

// ------------------------------------------------------------------------------------------
// 256x128 software-pipelined kernel (large M): 512 threads = 8 waves (4 along M x 2 along N) of
// 64x64, one workgroup per CU.  Same DMA staging / swizzle / slab reuse as the kernel above, plus
//  * half the weight traffic per FLOP (each 128x32 weight tile feeds 256 frames);
//  * a 3-deep LDS ring for the weight tiles (and for the slab when w == 1): the DMA for step s+2
//    is issued in step s, so a tile has a whole step to land before anyone reads it;
//  * register double-buffered fragments: the 16 ds_read_b128 of step s+1 are issued BEFORE the
//    24 MFMAs of step s, so the MFMA stream of a wave never waits on LDS.
// Ablation on the 128x128 kernel (profiles/): ds_read + MFMA alone run at ~87 % of the clock-
// adjusted MFMA peak; DMA traffic costs 24 % and the barrier 4 % of the loop.
namespace {
constexpr int PBM = 256, PNT = 512;
constexpr int PA_ROWS = PBM + 8;                   // slab rows incl. w-1 <= 7 halo rows
constexpr int PA_BYTES = PA_ROWS * DROW;           // 33792
constexpr int PB_BYTES = BN * DROW;                // 16384
struct Frags { bf16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2]; };   // [k16 step][tile]
}  // namespace

__global__ __launch_bounds__(512, 2) void gemm_bf16x3_pipe_kernel(GemmArgs p, int nMt, int nNt, int w) {
  extern __shared__ __attribute__((aligned(16))) char smem3[];
  char* As = smem3;                      // [3][PA_ROWS][128]
  char* Bs = smem3 + 3 * PA_BYTES;       // [3][BN][128]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;

  const int tile = xcd_remap(blockIdx.x, nMt * nNt);
  const int mt = tile / nNt, nt = tile - mt * nNt;
  const int m0 = mt * PBM, n0 = nt * BN;

  const int ncb = (p.Kpad >> 5) / w;               // channel blocks per frame
  const int nsteps = ncb * w;
  const int ngroups = (PBM + w - 1 + 7) >> 3;       // 8-row DMA groups per slab (32 or 33)
  const int aring = (w == 1) ? 3 : 2;               // slab ring depth
  const int gps = (w == 1) ? ngroups : (ngroups + w - 2) / (w - 1);   // slab groups issued per step
  const int lrow = lane >> 3, lpc = lane & 7;
  const int64_t a_row_bytes = p.ldsbx * 4, b_row_bytes = (int64_t)p.Kpad * 4;
  const char* Ag = reinterpret_cast<const char*>(p.Xsb) + (int64_t)(m0 + lrow) * a_row_bytes;
  const char* Bg = reinterpret_cast<const char*>(p.Wsb) + (int64_t)(n0 + lrow) * b_row_bytes;

  auto dma_a = [&](int cb, int buf, int g) {
    const int c = lpc ^ ((4 * g + (lrow >> 1)) & 7);
    __builtin_amdgcn_global_load_lds((gptr_t)(Ag + (int64_t)(8 * g) * a_row_bytes + cb * 128 + c * 16),
                                     (lptr_t)(As + buf * PA_BYTES + g * 1024), 16, 0, 0);
  };
  auto dma_b = [&](int kb, int buf, int g) {
    const int c = lpc ^ ((4 * g + (lrow >> 1)) & 7);
    __builtin_amdgcn_global_load_lds((gptr_t)(Bg + (int64_t)(8 * g) * b_row_bytes + (int64_t)kb * 128 + c * 16),
                                     (lptr_t)(Bs + buf * PB_BYTES + g * 1024), 16, 0, 0);
  };

  // B fragment offsets (row fixed per lane): chunk c = plane*4 + ks*2 + h
  int boff[2], bswz[2];
#pragma unroll
  for (int nj = 0; nj < 2; ++nj) {
    const int rb = wn * 64 + nj * 32 + r32;
    boff[nj] = rb * DROW;
    bswz[nj] = (rb >> 1) & 7;
  }
  auto read_frags = [&](Frags& f, int abuf, int j, int bbuf) {
    const char* ab = As + abuf * PA_BYTES;
    const char* bb = Bs + bbuf * PB_BYTES;
    int aoff[2], aswz[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int ra = wm * 64 + mi * 32 + r32 + j;
      aoff[mi] = ra * DROW;
      aswz[mi] = (ra >> 1) & 7;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ch = ks * 2 + h, cl = 4 + ks * 2 + h;
        f.ah[ks][i] = *reinterpret_cast<const bf16x8*>(ab + aoff[i] + ((ch ^ aswz[i]) << 4));
        f.al[ks][i] = *reinterpret_cast<const bf16x8*>(ab + aoff[i] + ((cl ^ aswz[i]) << 4));
        f.bh[ks][i] = *reinterpret_cast<const bf16x8*>(bb + boff[i] + ((ch ^ bswz[i]) << 4));
        f.bl[ks][i] = *reinterpret_cast<const bf16x8*>(bb + boff[i] + ((cl ^ bswz[i]) << 4));
      }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  auto mfma24 = [&](const Frags& f) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
          acc[nj][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.bh[ks][nj], f.al[ks][mi], acc[nj][mi], 0, 0, 0);
          acc[nj][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.bl[ks][nj], f.ah[ks][mi], acc[nj][mi], 0, 0, 0);
          acc[nj][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.bh[ks][nj], f.ah[ks][mi], acc[nj][mi], 0, 0, 0);
        }
  };

  // (cb, j) of steps s, s+1, s+2 and their ring slots, advanced incrementally
  int cb0 = 0, j0 = 0, cb1 = 0, j1 = 1, cb2 = 0, j2 = 2;
  auto norm = [&](int& cb, int& j) { while (j >= w) { j -= w; ++cb; } };
  norm(cb1, j1);
  norm(cb2, j2);
  int b0 = 0, b1 = 1, b2 = 2;            // weight ring slots of steps s, s+1, s+2

  // prologue: slab 0 (and slab 1 when w == 1), weight tiles of steps 0 and 1
  for (int g = wave; g < ngroups; g += 8) dma_a(0, 0, g);
  if (w == 1 && ncb > 1)
    for (int g = wave; g < ngroups; g += 8) dma_a(1, 1, g);
#pragma unroll
  for (int q = 0; q < 2; ++q) dma_b(0, 0, wave + 8 * q);
  if (nsteps > 1) {
#pragma unroll
    for (int q = 0; q < 2; ++q) dma_b(j1 * ncb + cb1, 1, wave + 8 * q);
  }
  __syncthreads();

  Frags f0, f1;
  read_frags(f0, 0, 0, 0);

  auto body = [&](int s, const Frags& fc, Frags& fn) {
    __builtin_amdgcn_s_waitcnt(0xc07f);             // lgkmcnt(0): fragments of step s are in registers
    // DMA two steps ahead: weight tile of step s+2, and this step's share of the next slab
    if (s + 2 < nsteps) {
#pragma unroll
      for (int q = 0; q < 2; ++q) dma_b(j2 * ncb + cb2, b2, wave + 8 * q);
    }
    if (w == 1) {
      if (cb0 + 2 < ncb)
        for (int g = wave; g < ngroups; g += 8) dma_a(cb0 + 2, (cb0 + 2) % 3, g);
    } else if (cb0 + 1 < ncb && j0 < w - 1) {
      const int gend = min((j0 + 1) * gps, ngroups);
      for (int g = j0 * gps + wave; g < gend; g += 8) dma_a(cb0 + 1, (cb0 + 1) & 1, g);
    }
    // fragments of step s+1 (their tiles landed before the previous barrier)
    if (s + 1 < nsteps) read_frags(fn, aring == 3 ? cb1 % 3 : (cb1 & 1), j1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mfma24(fc);
    __syncthreads();
    cb0 = cb1; j0 = j1; cb1 = cb2; j1 = j2;
    ++j2; if (j2 == w) { j2 = 0; ++cb2; }
    const int t = b0; b0 = b1; b1 = b2; b2 = t;
  };

  for (int s = 0; s < nsteps; s += 2) {
    body(s, f0, f1);
    if (s + 1 < nsteps) body(s + 1, f1, f0);
  }

  // the final barrier of the K loop has retired every LDS read: reuse the ring as store scratch
  store_wave_tile(p, acc, m0 + wm * 64, n0 + wn * 64, lane, wave, smem3);
}

hipError_t launch_gemm_bf16x3(const GemmArgs& a, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  static int force = -1;        // XVEC_GEMM_TILE=128 register-staged | 1 DMA 128x128 | 256 pipelined 256x128; 0 = by size
  static bool attr_set = false;
  static int diag = 0;              // XVEC_GEMM_DIAG: timing-only ablation switches (outputs invalid)
  const size_t smem128 = (size_t)4 * TILE_B;
  const size_t smemdma = (size_t)2 * DA_BYTES + 2 * DB_BYTES;
  const size_t smempipe = (size_t)3 * PA_BYTES + 3 * PB_BYTES;
  if (!attr_set) {
    const char* e = getenv("XVEC_GEMM_TILE");
    force = e ? atoi(e) : 0;
    hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem128);
    if (r != hipSuccess) return r;
    r = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_dma_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smemdma);
    if (r != hipSuccess) return r;
    r = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_pipe_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smempipe);
    if (r != hipSuccess) return r;
    const char* e3 = getenv("XVEC_GEMM_DIAG");
    diag = e3 ? atoi(e3) : 0;
    attr_set = true;
  }
  const int w = a.K / a.cin > 0 && a.ldsbx == a.cin ? a.K / a.cin : 1;   // taps (dense: 1)
  const bool taps_ok = w <= 9 && (a.Kpad >> 5) % w == 0;   // slab halo: 128 + w - 1 <= DA_ROWS (136)
  if (taps_ok && w <= 8 && force == 256) {   // A/B variant only: measured slower than the 128x128 DMA kernel (profiles/)
    const int nMt = (a.M + PBM - 1) / PBM, nNt = a.Npad / BN;
    hipLaunchKernelGGL(gemm_bf16x3_pipe_kernel, dim3(nMt * nNt), dim3(PNT), smempipe, s, a, nMt, nNt, w);
    return hipGetLastError();
  }
  if (force != 128 && taps_ok) {
    const int nMt = (a.M + BM - 1) / BM, nNt = a.Npad / BN;
    hipLaunchKernelGGL(gemm_bf16x3_dma_kernel, dim3(nMt * nNt), dim3(256), smemdma, s, a, nMt, nNt, w, diag);
    return hipGetLastError();
  }
  const int nMt = (a.M + BM - 1) / BM, nNt = a.Npad / BN;
  hipLaunchKernelGGL(gemm_bf16x3_kernel, dim3(nMt * nNt), dim3(256), smem128, s, a, nMt, nNt);
  return hipGetLastError();
}

}  // namespace xv
