// bf16x3 split-precision MFMA GEMM for gfx950 (MI355X): the TDNN's temporal convolutions and
// dense layers at ~fp32 accuracy on the bf16 matrix pipe.
//
//   x = hi + lo (+ O(2^-17 |x|)),  hi = rn_bf16(x), lo = rn_bf16(x - hi)
//   a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (dropped terms ~2^-17 |a b|)
//
// Three v_mfma_f32_16x16x32_bf16 (or _f16: XV_PREC_F16X3) per product tile, fp32 accumulate: 16/3 = 5.3x the rate of
// the exact v_mfma_f32_32x32x2_f32 path at a relative error of ~5e-6 per layer (measured in
// tests/test_gpu_parity.py against the float64 oracle; the bar is 1e-4).  The lab kernels below still use the
// 32x32x16 shape, on which the chip holds a lower clock (xv_epilogue.h, mfma_split16).
//
// Both operands arrive in the split-blocked format of xv_epilogue.h: per row, 128-byte blocks
// [32 x bf16 hi | 32 x bf16 lo].  One block is one K step of 32 for one row, so
//   * every global access is a full 128-byte line (8 lanes x 16 bytes),
//   * 16-byte chunk q of a block is exactly a lane's MFMA fragment: plane q>>2 (hi / lo), k chunk q&3 = lane>>4 of
//     the 16x16x32 shape (k16-step (q>>1)&1, lane-half q&1 of 32x32x16) -- one ds_read_b128 per fragment, no repacking,
//   * a temporal convolution stays an "overlapping-row" GEMM: row m of A is the contiguous
//     run of w*cin/32 blocks that starts at frame m.
//
// The product kernels, on 128x128 workgroup tiles of 256 threads:
//   gemm_bf16x3_w14p2_kernel  multi-tap convolutions: activation slabs by LDS-DMA (two buffers, one slab per channel
//                             block), weight fragments straight into registers two steps ahead, one wave per
//                             32-channel block, counted waits; the K-split slices of the tail tiles ride in the same launch;
//   gemm_bf16x3_w1p3_kernel   one-tap layers (dense, attention epilogues, ResNet grid convolutions incl. the gathered
//                             compact-row form): three slab buffers, slabs two steps ahead.
// Lab builds only (-DXV_LAB; tools/ab_variants.sh -- never in the shipped library):
//   gemm_bf16x3_dma_kernel    its predecessor, the A/B baseline (XVEC_GEMM_TILE=1) and the vehicle of the timing
//                             ablations (XVEC_GEMM_DIAG, which produce wrong results on purpose): both operands by
//                             LDS-DMA, one barrier per K step;
//   gemm_bf16x3_kernel        register-staged form (XVEC_GEMM_TILE=128): LDS rows padded to 144 bytes, double
//                             buffering, one barrier per K step, 2x2 waves of 64x64, two workgroups per CU.
#include <cstdlib>
#include <mutex>

#include "xv_f6.h"

namespace xv {

namespace {
constexpr int BM = 128, BN = 128;
constexpr int ROWB = 144;                 // padded LDS bytes per (row, K-step)
constexpr int TILE_B = BM * ROWB;         // bytes per operand tile
typedef int i32x4 __attribute__((ext_vector_type(4)));
}  // namespace

namespace {
constexpr int DROW = 128;                          // unpadded LDS row
constexpr int DA_ROWS = 136;                       // 128 + (w-1 <= 8) halo rows, multiple of 8
constexpr int DA_BYTES = DA_ROWS * DROW;
constexpr int DB_BYTES = BN * DROW;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
}  // namespace

#ifdef XV_LAB
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
// 128x128 LDS-DMA kernel (the round's first default, now the A/B baseline XVEC_GEMM_TILE=1): same 2x2 waves of
// 64x64, two workgroups per CU, but
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
        bh[ks][i] = *reinterpret_cast<const bf16x8*>(bb + boff[i] + ((ch ^ bswz[i]) << 4));
        if (diag & 64) {                 // diag bit6: half the LDS fragment reads (timing only, wrong results)
          al[ks][i] = ah[ks][i];
          bl[ks][i] = bh[ks][i];
        } else {
          al[ks][i] = *reinterpret_cast<const bf16x8*>(ab + aoff[i] + ((cl ^ aswz[i]) << 4));
          bl[ks][i] = *reinterpret_cast<const bf16x8*>(bb + boff[i] + ((cl ^ bswz[i]) << 4));
        }
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

#endif  // XV_LAB

// ------------------------------------------------------------------------------------------
// Default kernel: 128x128 workgroup tile, "weights in registers", 1 x 4 wave layout, counted waits.
//
// How it got here (each step measured on MI355X, profiles/README.md; the intermediate kernels are in the git
// history: 3fd7543 weights in registers 2 x 2 waves, 659ee67 counted waits, c74e274 1 x 4 waves):
//  * the activation slab is staged exactly as in the LDS-DMA kernel above, but the weight fragments never touch
//    LDS: they are packed a second time in MFMA-fragment-major order (one global_load_dwordx4 of a wave = 1 KB
//    contiguous; reading the row-major SB rows directly touches 32 lines per instruction and is 60 % slower) and
//    every lane loads its own fragments straight from L2 into VGPR buffers.  85 % of the LDS-DMA pieces of the
//    kernel above were weight tiles; they and their landing wait go, and the only LDS hazard left is the slab,
//    so the workgroup barrier drops from one per K step to one per channel block (every w steps);
//  * the epilogue scratch is 8 KB per wave (64 frames x 128 B per pass), the workgroup needs only the 34 KB of
//    the slabs, and THREE workgroups share a CU (12 waves);
//  * every wave owns ONE 32-channel block and all 128 frames of the tile.  With 2 x 2 waves the two waves of a
//    pair load identical weight fragments; the L1 counters showed the second request landing on a line still in
//    flight 36 % of all L1 cycles (TCP_PENDING_STALL_CYCLES, profiles/r01/pmc_tcp1_wreg.txt) and the texture
//    addresser busy 62 % of the kernel.  Here every weight byte is requested once per workgroup (4 loads per
//    wave and step); the price is that all four waves read the whole slab from LDS (16 ds_read_b128 per wave and
//    step), which the LDS has room for (halving the LDS reads of the 2 x 2 kernel changed nothing);
//  * every vector-memory instruction of the K loop is inline assembly, issued one per group of 3 MFMAs in the
//    order the fragments are needed (a burst from all waves of the CU queues up in the texture addresser and the
//    in-order waves stop feeding the matrix pipe: -4..7 %), and every wait is an explicit s_waitcnt vmcnt(N) with
//    an exact count.  The compiler can only drain the counter (vmcnt(0)) when LDS-DMA and register loads are
//    mixed, and it needs the issue to be unconditional (a path that skips a load forces vmcnt(0)), hence the
//    clamped duplicate slab pieces instead of branches (NPS pieces per wave and step: 1 when w >= 5, else 4).
//    The counts rely on LDS-DMA loads and register loads retiring in issue order under the one counter (checked
//    on hardware: tools/vmcnt_order_test.hip, 0 violations in 2.6e8 trials).  The waits name the registers they
//    release as "+v" operands, so the compiler cannot move an MFMA above its wait.
#define XV_GLD(dst, ptr, OFF) asm volatile("global_load_dwordx4 %0, %1, off offset:" #OFF : "=v"(dst) : "v"(ptr))
#define XV_WAIT2(N, ra, rb) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(ra), "+v"(rb) : "n"(N))
#define XV_WAIT4(N, ra, rb, rc, rd) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ra), "+v"(rb), "+v"(rc), "+v"(rd) : "n"(N))

//  * the weight fragments are prefetched TWO steps ahead (three register buffers of 4 fragments).
//    With one step of lead a workgroup alone on a CU is bound by the load->use latency (~0.9 us per step against
// 0.4 us of MFMA work), so only three resident workgroups together cover the matrix pipe and every prologue,
// epilogue and the tail of the launch leave it under-fed; with two steps of lead two workgroups suffice.
//   VMEM order per step:  D x NPS (top) , a0 (group 0) a1 (group 2) a2 (group 4) a3 (group 6)   [for step s+2]
//   before the first MFMA of step s: vmcnt(5+2*NPS) [a0..a3 of step s, issued during step s-2: younger are the NPS slab
//   pieces + 4 weight loads of step s-1 and the NPS pieces + a0 of step s]; slab switch vmcnt(4).
//  * MFMA shape v_mfma_f32_16x16x32 (round 2): the wave tile is 8 x 2 accumulator tiles of 16 frames x 16 channels; per K
//    step 48 MFMAs of 16 cycles instead of 24 of 32 -- the same operand bytes, LDS reads and FLOP per cycle, but the chip
//    holds a clearly higher clock on this shape (a timing-only swap of the instruction inside this kernel: tdnn3_conv
//    0.591 -> 0.500 ms; profiles/README.md).
//  * K-split slices: the tile function takes the channel-block range [cb_begin, cb_end) so that the tail form
//    (gemm_bf16x3_tail_plan) can run one slice of K per workgroup.
// The activation fragment addresses use ONE swizzle and ONE row offset per lane ((32*mi + r) >> 1 == r >> 1 mod 8)
// plus ds_read immediates, which frees the registers for the third weight buffer at 3 workgroups / CU.
// EPI selects the epilogue at compile time: 0 = activations (fp32 / SB / residual / fused statistics pooling),
// 1 / 2 = the attention forms of xv_epilogue.h (score partials of the last key layer / weighted moments of the value).
// A separate instantiation keeps the default kernel's code -- and its register allocation -- untouched.
// F16 selects the split format of both operands (bf16 hi/lo or fp16 hi/lo: same bytes, same MFMA rate).
// Traffic lab (-DXV_TRAFFIC_LAB=bits, tools/traffic_lab.sh; never in the shipped library; results are WRONG with bits 0/1):
//   bit 0  every workgroup stages the activation slab of M tile 0      -> FETCH_SIZE shows the weight stream alone
//   bit 1  every workgroup loads the weight fragments of N tile 0      -> FETCH_SIZE shows the activation stream alone
//   bit 2  tile order: one N tile per XCD (weights of an XCD stay in its L2; every M tile is staged on nNt XCDs)
//   bit 3  tile order: an XCD walks its M range once per N tile (N-major phases)
//   bit 4  tile order: two N tiles per XCD (half of the weights per L2; every M tile is staged on two XCDs)
#ifndef XV_TRAFFIC_LAB
#define XV_TRAFFIC_LAB 0
#endif
#if (XV_TRAFFIC_LAB & 32)
#define XV_TLAB_ANT " nt"      // bit 5: slab loads non-temporal (evict-first in L2)
#else
#define XV_TLAB_ANT ""
#endif
#define XV_TLAB_M0(m0) ((XV_TRAFFIC_LAB & 1) ? 0 : (m0))
#define XV_TLAB_N0(n0) ((XV_TRAFFIC_LAB & 2) ? 0 : (n0))

// workgroup -> (M tile, N tile).  Product order: ids congruent mod 8 share an XCD; an XCD owns a contiguous run of
// tiles, N fastest, so the nNt workgroups that stage the same slab start together on the same L2.
__device__ __forceinline__ void tile_of_block(int id, int nMt, int nNt, int& mt, int& nt) {
  if constexpr ((XV_TRAFFIC_LAB & 4) != 0) {
    if (nNt == 4) {                                   // XCD x: N tile x & 3, M tiles of half x >> 2
      const int x = id & 7, i = id >> 3, h0 = (nMt + 1) >> 1;
      nt = x & 3;
      mt = (x >> 2) ? h0 + i : i;
      return;
    }
  }
  if constexpr ((XV_TRAFFIC_LAB & 8) != 0) {
    if ((nMt & 7) == 0) {                             // XCD x: M tiles [x * nMt/8, (x+1) * nMt/8), N-major
      const int x = id & 7, i = id >> 3, per = nMt >> 3;
      nt = i / per;
      mt = x * per + (i - nt * per);
      return;
    }
  }
  if constexpr ((XV_TRAFFIC_LAB & 16) != 0) {
    if (nNt == 4 && (nMt & 3) == 0) {                 // XCD x: N tiles 2(x&1), 2(x&1)+1; M quarter x >> 1
      const int x = id & 7, i = id >> 3, per = nMt >> 2;
      nt = 2 * (x & 1) + (i & 1);
      mt = (x >> 1) * per + (i >> 1);
      return;
    }
  }
  const int tile = xcd_remap(id, nMt * nNt);
  mt = tile / nNt;
  nt = tile - mt * nNt;
}

template <int NPS, int EPI = 0, bool F16 = false>
__device__ __forceinline__ void w14p2_tile(const GemmArgs& p, int m0, int n0, int w, char* smem3, int cb_begin, int cb_end, bool slice = false) {
  char* As = smem3;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g4 = lane >> 4;       // 16x16x32 operand maps: row / column lane & 15, k chunk lane >> 4

  const int ncb = (p.Kpad >> 5) / w;
  const int nsteps = (cb_end - cb_begin) * w;      // channel blocks [cb_begin, cb_end): the whole K, or one K-split slice
  const int ngroups = (BM + w - 1 + 7) >> 3;
  const int lrow = lane >> 3, lpc = lane & 7;
  const int64_t a_row_bytes = (p.a_pitch ? p.a_pitch : p.ldsbx) * 4;
  const int kbt = p.a_pitch ? (p.ktap >> 5) : (p.Kpad >> 5);
  const int64_t tap_bytes = p.a_pitch ? p.tap_stride * 4 : 0;
  const char* Ag = reinterpret_cast<const char*>(p.Xsb) + p.a_off * 4 + (int64_t)(XV_TLAB_M0(m0) + lrow) * a_row_bytes;
  const int64_t nkb4k = (int64_t)(p.Kpad >> 5) * 4096;
  const char* Wg = reinterpret_cast<const char*>(p.Wfr) + (int64_t)((XV_TLAB_N0(n0) >> 5) + wave) * nkb4k + lane * 16;
  const uint32_t as_lds = (uint32_t)(uintptr_t)(lptr_t)As;

  auto dma_a = [&](int64_t koff, int buf, int g) {
    const int c = lpc ^ (((lrow >> 1) & 3) << 1);          // slab swizzle (below): 2 * ((row >> 1) & 3), row = 8 g + lrow
    const char* src = Ag + (int64_t)(8 * g) * a_row_bytes + koff + c * 16;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(as_lds + buf * DA_BYTES + g * 1024);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" XV_TLAB_ANT ::"v"(src), "s"(dst) : "memory");
  };

  f32x4 acc[8][2];                       // [16-frame tile][16-channel tile]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // (cb, j) of the current step, of the next one and of the one after
  int cb = cb_begin, j = 0, cb1 = cb_begin, j1 = 1, cb2, j2;
  if (j1 == w) { j1 = 0; cb1 = cb_begin + 1; }
  cb2 = cb1; j2 = j1 + 1;
  if (j2 == w) { j2 = 0; cb2 = cb1 + 1; }

  auto stamp = [&](int i) {
#ifdef XV_GEMM_TRACE
    if (p.trace && tid == 0) p.trace[(int64_t)blockIdx.x * 8 + i] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  };
  stamp(0);
  bf16x8 W0[4], W1[4], W2[4];            // [plane * 2 + channel tile]; W0: step 0, W1: step 1
  {
    const char* q1 = Wg + (int64_t)(1 < nsteps ? j1 * ncb + cb1 : 0) * 4096;
    const char* q0 = Wg + (int64_t)cb_begin * 4096;               // step 0 = (cb_begin, tap 0)
    XV_GLD(W0[0], q0, 0); XV_GLD(W0[1], q0, 1024); XV_GLD(W0[2], q0, 2048); XV_GLD(W0[3], q0, 3072);
    XV_GLD(W1[0], q1, 0); XV_GLD(W1[1], q1, 1024); XV_GLD(W1[2], q1, 2048); XV_GLD(W1[3], q1, 3072);
  }
  // (a K-split slice starts at block cb_begin of the row; slices exist in the 1-D form only)
  for (int g = wave; g < ngroups; g += 4) dma_a((int64_t)cb_begin * 128, cb_begin & 1, g);
  int64_t koff_next = (kbt == 1 && p.a_pitch) ? tap_bytes : (int64_t)(cb_begin + 1) * 128;
  int blk_next = (kbt == 1 && p.a_pitch) ? 0 : 1;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  stamp(1);
  auto step = [&](int s, bf16x8 (&Wc)[4], bf16x8 (&Wn)[4]) __attribute__((always_inline)) {
    const int kbn = s + 2 < nsteps ? j2 * ncb + cb2 : 0;        // weights of step s + 2 (unconditional issue)
    const char* q = Wg + (int64_t)kbn * 4096;
    const char* ab = As + (cb & 1) * DA_BYTES + (c16 + j) * DROW;
    // Slab swizzle: chunk c of slab row r lives at position c ^ 2 * ((r >> 1) & 3).  A ds_read_b128 is served in groups of 16
    // lanes = 16 consecutive rows of which the outer eight read k chunk g4 and the inner eight g4 ^ 1 (lane groups of the
    // instruction: MI355X_MICROARCH.md, LDS); with the tap shift j the window starts at any row, and only a swizzle that leaves
    // bit 0 of the chunk alone keeps those two sets apart for every start (the round-2 swizzle (r >> 1) & 7 was conflict-free
    // for j = 0 mod 4 only: tests/analysis/lds_bank_model.py, SQ_LDS_BANK_CONFLICT).  (16 ft + r) >> 1 == r >> 1 (mod 4).
    const int aswz = (((c16 + j) >> 1) & 3) << 1;
    const int off_hi = (g4 ^ aswz) << 4, off_lo = ((4 + g4) ^ aswz) << 4;     // this lane's k chunk of the hi / lo half
    bf16x8 fh[8], fl[8];                 // activation fragments of frame tile ft, read two tiles ahead
    auto read_frag = [&](int ft) {
      fh[ft] = *reinterpret_cast<const bf16x8*>(ab + off_hi + ft * (16 * DROW));
      fl[ft] = *reinterpret_cast<const bf16x8*>(ab + off_lo + ft * (16 * DROW));
    };
    read_frag(0);
    read_frag(1);
    {
      const int64_t ksrc = cb + 1 < cb_end ? koff_next : 0;
#pragma unroll
      for (int i = 0; i < NPS; ++i) dma_a(ksrc, (cb + 1) & 1, min((j * NPS + i) * 4 + wave, ngroups - 1));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 8; ++g) {        // g = frame tile: 6 MFMAs of 16 cycles
      if (g == 0) XV_GLD(Wn[0], q, 0);        // hi, channels 0..15
      if (g == 2) XV_GLD(Wn[1], q, 1024);     // hi, channels 16..31
      if (g == 4) XV_GLD(Wn[2], q, 2048);     // lo, channels 0..15
      if (g == 6) XV_GLD(Wn[3], q, 3072);     // lo, channels 16..31
      if (g + 2 < 8) read_frag(g + 2);
      if (g == 0) XV_WAIT4(5 + 2 * NPS, Wc[0], Wc[1], Wc[2], Wc[3]);
      // small cross terms first, the dominant hi*hi term last; the two channel tiles alternate
#ifdef XV_F6_TIMING
      // TIMING ONLY (wrong results): the instruction mix of the proposed two-unit split on this kernel's
      // memory traffic -- per K step the hi*hi MFMA of each tile, and every fourth step the two block-scaled fp6 cross terms of
      // four steps, with whatever bits the operand registers hold
      acc[g][0] = mfma_split16<true>(Wc[0], fh[g], acc[g][0]);
      acc[g][1] = mfma_split16<true>(Wc[1], fh[g], acc[g][1]);
      if ((s & 3) == 3) {
        typedef int v8i_t __attribute__((ext_vector_type(8)));
        typedef int v4i_t __attribute__((ext_vector_type(4)));
        const v4i_t w0 = __builtin_bit_cast(v4i_t, Wc[0]), w1 = __builtin_bit_cast(v4i_t, Wc[1]), w2 = __builtin_bit_cast(v4i_t, Wc[2]),
                    w3 = __builtin_bit_cast(v4i_t, Wc[3]), b0 = __builtin_bit_cast(v4i_t, fh[g]), b1 = __builtin_bit_cast(v4i_t, fl[g]);
        const v8i_t a0 = {w0[0], w0[1], w0[2], w0[3], w2[0], w2[1], w2[2], w2[3]}, a1 = {w1[0], w1[1], w1[2], w1[3], w3[0], w3[1], w3[2], w3[3]};
        const v8i_t bb = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        acc[g][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a0, bb, acc[g][0], 2, 2, 0, 127, 0, 127);
        acc[g][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a1, bb, acc[g][1], 2, 2, 0, 127, 0, 127);
        acc[g][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a1, bb, acc[g][0], 2, 2, 0, 127, 0, 127);
        acc[g][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a0, bb, acc[g][1], 2, 2, 0, 127, 0, 127);
      }
#else
      acc[g][0] = mfma_split16<F16>(Wc[0], fl[g], acc[g][0]);
      acc[g][1] = mfma_split16<F16>(Wc[1], fl[g], acc[g][1]);
      acc[g][0] = mfma_split16<F16>(Wc[2], fh[g], acc[g][0]);
      acc[g][1] = mfma_split16<F16>(Wc[3], fh[g], acc[g][1]);
      acc[g][0] = mfma_split16<F16>(Wc[0], fh[g], acc[g][0]);
      acc[g][1] = mfma_split16<F16>(Wc[1], fh[g], acc[g][1]);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    if (cb1 != cb) {                     // slab switch
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __syncthreads();
      if (++blk_next == kbt) {
        blk_next = 0;
        koff_next += tap_bytes - (int64_t)(kbt - 1) * 128;
      } else {
        koff_next += 128;
      }
    }
    cb = cb1; j = j1;
    cb1 = cb2; j1 = j2;
    if (++j2 == w) { j2 = 0; ++cb2; }
  };
  for (int s = 0; s < nsteps; s += 3) {
    step(s, W0, W2);
    if (s + 1 < nsteps) step(s + 1, W1, W0);
    if (s + 2 < nsteps) step(s + 2, W2, W1);
  }
  // every wave's last (dummy) slab pieces have landed before any wave turns the slab buffers into its epilogue scratch:
  // the waves of a workgroup can be a step apart here, and a piece issued at the top of that step may still be in flight
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
  stamp(2);
  // the epilogue's per-lane addresses all derive from the lane id: made opaque here, they cannot be hoisted above the K
  // loop, where they would cost registers (and spill) for its whole duration
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  if constexpr (EPI == 3) {                // XV_PREC_F16F6 producer: the two-unit block format
    if (slice) {
      // K-split slice of a tail tile: the raw accumulators go to the slice's rows of GemmArgs::partial (p.Y / p.ldy were pointed
      // there by the kernel), 16-byte stores straight from the accumulator layout as in gemm_f6v2.hip -- a second LDS epilogue in
      // this instantiation would not fit its registers
      const int c16e = lane_e & 15, g4e = lane_e >> 4;
#pragma unroll
      for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          *reinterpret_cast<f32x4*>(p.Y + (int64_t)(m0 + 16 * g + c16e) * p.ldy + n0 + wave * 32 + 16 * c + 4 * g4e) = acc[g][c];
      return;
    }
    store_wave_tile_n32_f6(p, acc, m0, n0 + wave * 32, lane_e, wave, smem3);
  } else if constexpr (EPI != 0) store_wave_tile_n32_att<EPI>(p, acc, m0, n0 + wave * 32, lane_e, wave, smem3);
  else store_wave_tile_n32<64, F16>(p, acc, m0, n0 + wave * 32, lane_e, wave, smem3);
  stamp(3);
}

// One-tap form (dense layers, and the ResNet grid convolutions whose taps are plain K blocks): every K step needs a new
// activation slab.  With two slab buffers the slab of step s + 1 is issued at the top of step s and has to land within
// that one step (the wait at the end of the step exposes whatever is left of its ~1 us LDS-DMA latency, every step):
// measured 1.46-1.75 us per step against 1.36 us for the 7-tap convolution, whose slab has seven steps to land.  Here the
// slabs rotate through THREE buffers (52 KB of LDS, still three workgroups per CU) and are issued TWO steps ahead, like
// the weights; the wait at the end of step s only has to retire slab s + 1, issued a whole step earlier:
//   VMEM order per step:  D x 4 (slab s+2, top) , a0 a1 a2 a3 (weights s+2)
//   before the first MFMA: vmcnt(13) [younger than the weights of step s: slab s+1 (4), weights s+1 (4), slab s+2 (4),
//   the first weight load of s+2 (1)]; end of step: vmcnt(12)
//   [younger than slab s+1: weights s+1 (4), slab s+2 (4), weights s+2 (4)], then the workgroup barrier (RAW for the slab
//   read in step s+1; WAR for buffer (s+3) % 3 = s % 3, refilled at the top of step s+1).
// The loop is unrolled by three, so the slab buffer of a step is a compile-time constant like its weight registers.
template <int EPI, bool F16, bool GATHER = false>
__device__ __forceinline__ void w1p3_tile(const GemmArgs& p, int m0, int n0, char* smem3) {
  char* As = smem3;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g4 = lane >> 4;       // 16x16x32 operand maps: row / column lane & 15, k chunk lane >> 4

  const int nsteps = p.Kpad >> 5;
  const int lrow = lane >> 3, lpc = lane & 7;
  const int64_t a_row_bytes = (p.a_pitch ? p.a_pitch : p.ldsbx) * 4;
  const int kbt = p.a_pitch ? (p.ktap >> 5) : (p.Kpad >> 5);
  const int64_t tap_bytes = p.a_pitch ? p.tap_stride * 4 : 0;
  const char* Ag = reinterpret_cast<const char*>(p.Xsb) + p.a_off * 4 + (int64_t)(XV_TLAB_M0(m0) + lrow) * a_row_bytes;
  const int64_t nkb4k = (int64_t)(p.Kpad >> 5) * 4096;
  // Channel-block role of this wave.  A layer narrower than the tile (ResNet stage 1: 64 channels in a 128-channel tile)
  // leaves two of the four blocks pure padding: the waves that own them only stage their share of the slabs and keep the
  // barriers (no weight loads, no MFMAs, no epilogue), and WHICH waves those are alternates with the workgroup's dispatch
  // slot, so that the computing waves of the workgroups sharing a CU do not all sit on the same two SIMDs.
  const bool narrow = EPI == 0 && p.N <= 64 && !p.pool_part && !p.raw;
  const int wv = narrow ? (wave ^ (((blockIdx.x >> 8) & 1) << 1)) : wave;
  const bool passive = narrow && n0 + wv * 32 >= p.N;
  const char* Wg = reinterpret_cast<const char*>(p.Wfr) + (int64_t)((XV_TLAB_N0(n0) >> 5) + wv) * nkb4k + lane * 16;
  const uint32_t as_lds = (uint32_t)(uintptr_t)(lptr_t)As;

  // gathered form: this lane's four slab rows (groups wave, 4 + wave, 8 + wave, 12 + wave) start at grid positions
  // arow[...]; their byte offsets from the operand base (swizzled chunk included: (4 g) & 7 does not depend on i) stay in
  // four registers and every DMA is base (scalar, advanced with the K block) + offset (vector)
  uint32_t goff[4] = {0, 0, 0, 0};
  const char* gbase = reinterpret_cast<const char*>(p.Xsb) + p.a_off * 4;
  if constexpr (GATHER) {
    const int cg = lpc ^ ((4 * wave + (lrow >> 1)) & 7);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      goff[i] = (uint32_t)p.arow[m0 + lrow + 8 * (4 * i + wave)] * (uint32_t)a_row_bytes + cg * 16;
  }
  auto dma_a = [&](int64_t koff, int buf, int g) {
    const int c = lpc ^ ((4 * g + (lrow >> 1)) & 7);
    const char* src = Ag + (int64_t)(8 * g) * a_row_bytes + koff + c * 16;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(as_lds + buf * DA_BYTES + g * 1024);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" XV_TLAB_ANT ::"v"(src), "s"(dst) : "memory");
  };
  auto dma_g = [&](const char* kbase, int buf, int i) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane(as_lds + buf * DA_BYTES + (4 * i + wave) * 1024);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(goff[i]), "s"(kbase), "s"(dst) : "memory");
  };
  auto dma_slab = [&](int64_t koff, int buf) {      // the 16 eight-row groups of a slab: four per wave
    if constexpr (GATHER) {
      const char* kbase = gbase + koff;
#pragma unroll
      for (int i = 0; i < 4; ++i) dma_g(kbase, buf, i);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) dma_a(koff, buf, i * 4 + wave);
    }
  };

  f32x4 acc[8][2];                       // [16-frame tile][16-channel tile]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto stamp = [&](int i) {
#ifdef XV_GEMM_TRACE
    if (p.trace && tid == 0) p.trace[(int64_t)blockIdx.x * 8 + i] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  };
  stamp(0);
  // byte offset inside an A row of the K block of the slab to issue next (1-D form: block * 128; grid form: tap * tap_bytes
  // + block-in-tap * 128), advanced incrementally
  int64_t koff_issue = 0;
  int blk_issue = 0;
  auto advance = [&]() {
    if (++blk_issue == kbt) {
      blk_issue = 0;
      koff_issue += tap_bytes - (int64_t)(kbt - 1) * 128;
    } else {
      koff_issue += 128;
    }
  };
  if (passive) {                          // uniform per wave: slabs and barriers only (see `narrow` above)
    dma_slab(0, 0);
    advance();
    dma_slab(1 < nsteps ? koff_issue : 0, 1);
    advance();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = 0, buf = 2; s < nsteps; ++s) {
      dma_slab(s + 2 < nsteps ? koff_issue : 0, buf);        // slab s + 2 into buffer (s + 2) % 3
      buf = buf == 2 ? 0 : buf + 1;
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // this wave's pieces of slab s + 1 have landed
      __syncthreads();
      advance();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                      // the barrier in front of the other waves' epilogue (below)
    return;
  }
  bf16x8 W0[4], W1[4], W2[4];            // [plane * 2 + channel tile]; W0: step 0, W1: step 1
  {
    const char* q0 = Wg;
    const char* q1 = Wg + (int64_t)(1 < nsteps ? 1 : 0) * 4096;
    XV_GLD(W0[0], q0, 0); XV_GLD(W0[1], q0, 1024); XV_GLD(W0[2], q0, 2048); XV_GLD(W0[3], q0, 3072);
    XV_GLD(W1[0], q1, 0); XV_GLD(W1[1], q1, 1024); XV_GLD(W1[2], q1, 2048); XV_GLD(W1[3], q1, 3072);
  }
  dma_slab(0, 0);                        // slab 0
  advance();
  dma_slab(1 < nsteps ? koff_issue : 0, 1);          // slab 1 (a duplicate of slab 0 when K is a single block)
  advance();                             // koff_issue now describes slab 2
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  stamp(1);
  const int aswz = (c16 >> 1) & 7;       // (16 ft + r) >> 1 == r >> 1 (mod 8): one swizzle for all eight tiles
  const int off_hi = (g4 ^ aswz) << 4, off_lo = ((4 + g4) ^ aswz) << 4;       // this lane's k chunk of the hi / lo half
  auto step = [&](int s, int buf, bf16x8 (&Wc)[4], bf16x8 (&Wn)[4]) __attribute__((always_inline)) {
    const char* q = Wg + (int64_t)(s + 2 < nsteps ? s + 2 : 0) * 4096;      // weights of step s + 2 (unconditional issue)
    const char* ab = As + buf * DA_BYTES + c16 * DROW;
    bf16x8 fh[8], fl[8];                 // activation fragments of frame tile ft, read two tiles ahead
    auto read_frag = [&](int ft) {
      fh[ft] = *reinterpret_cast<const bf16x8*>(ab + off_hi + ft * (16 * DROW));
      fl[ft] = *reinterpret_cast<const bf16x8*>(ab + off_lo + ft * (16 * DROW));
    };
    read_frag(0);
    read_frag(1);
    dma_slab(s + 2 < nsteps ? koff_issue : 0, buf == 0 ? 2 : buf - 1);      // slab s + 2 into buffer (s + 2) % 3
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 8; ++g) {        // g = frame tile: 6 MFMAs of 16 cycles
      if (g == 0) XV_GLD(Wn[0], q, 0);        // hi, channels 0..15
      if (g == 2) XV_GLD(Wn[1], q, 1024);     // hi, channels 16..31
      if (g == 4) XV_GLD(Wn[2], q, 2048);     // lo, channels 0..15
      if (g == 6) XV_GLD(Wn[3], q, 3072);     // lo, channels 16..31
      if (g + 2 < 8) read_frag(g + 2);
      if (g == 0) XV_WAIT4(13, Wc[0], Wc[1], Wc[2], Wc[3]);
#ifdef XV_F6_TIMING
      // TIMING ONLY (wrong results): the instruction mix of the proposed two-unit split on this kernel's
      // memory traffic -- per K step the hi*hi MFMA of each tile, and every fourth step the two block-scaled fp6 cross terms of
      // four steps, with whatever bits the operand registers hold
      acc[g][0] = mfma_split16<true>(Wc[0], fh[g], acc[g][0]);
      acc[g][1] = mfma_split16<true>(Wc[1], fh[g], acc[g][1]);
      if ((s & 3) == 3) {
        typedef int v8i_t __attribute__((ext_vector_type(8)));
        typedef int v4i_t __attribute__((ext_vector_type(4)));
        const v4i_t w0 = __builtin_bit_cast(v4i_t, Wc[0]), w1 = __builtin_bit_cast(v4i_t, Wc[1]), w2 = __builtin_bit_cast(v4i_t, Wc[2]),
                    w3 = __builtin_bit_cast(v4i_t, Wc[3]), b0 = __builtin_bit_cast(v4i_t, fh[g]), b1 = __builtin_bit_cast(v4i_t, fl[g]);
        const v8i_t a0 = {w0[0], w0[1], w0[2], w0[3], w2[0], w2[1], w2[2], w2[3]}, a1 = {w1[0], w1[1], w1[2], w1[3], w3[0], w3[1], w3[2], w3[3]};
        const v8i_t bb = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        acc[g][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a0, bb, acc[g][0], 2, 2, 0, 127, 0, 127);
        acc[g][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a1, bb, acc[g][1], 2, 2, 0, 127, 0, 127);
        acc[g][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a1, bb, acc[g][0], 2, 2, 0, 127, 0, 127);
        acc[g][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a0, bb, acc[g][1], 2, 2, 0, 127, 0, 127);
      }
#else
      acc[g][0] = mfma_split16<F16>(Wc[0], fl[g], acc[g][0]);
      acc[g][1] = mfma_split16<F16>(Wc[1], fl[g], acc[g][1]);
      acc[g][0] = mfma_split16<F16>(Wc[2], fh[g], acc[g][0]);
      acc[g][1] = mfma_split16<F16>(Wc[3], fh[g], acc[g][1]);
      acc[g][0] = mfma_split16<F16>(Wc[0], fh[g], acc[g][0]);
      acc[g][1] = mfma_split16<F16>(Wc[1], fh[g], acc[g][1]);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // slab s + 1 has landed (this wave's pieces)
    __syncthreads();
    advance();
  };
  for (int s = 0; s < nsteps; s += 3) {
    step(s, 0, W0, W2);
    if (s + 1 < nsteps) step(s + 1, 1, W1, W0);
    if (s + 2 < nsteps) step(s + 2, 2, W2, W1);
  }
  // every wave's last (dummy) slab pieces have landed before any wave turns the slab buffers into its epilogue scratch:
  // the waves of a workgroup can be a step apart here, and a piece issued at the top of that step may still be in flight
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
  stamp(2);
  // the epilogue's per-lane addresses all derive from the lane id: made opaque here, they cannot be hoisted above the K
  // loop, where they would cost registers (and spill) for its whole duration
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  if constexpr (EPI == 3) store_wave_tile_n32_f6(p, acc, m0, n0 + wave * 32, lane_e, wave, smem3);      // XV_PREC_F16F6 producer
  else if constexpr (EPI != 0) store_wave_tile_n32_att<EPI>(p, acc, m0, n0 + wave * 32, lane_e, wave, smem3);
  else store_wave_tile_n32<64, F16>(p, acc, m0, n0 + wv * 32, lane_e, wave, smem3);
  stamp(3);
}

template <int EPI = 0, bool F16 = false, bool GATHER = false>
__global__ __launch_bounds__(256, 3) void gemm_bf16x3_w1p3_kernel(GemmArgs p, int nMt, int nNt) {
  extern __shared__ __attribute__((aligned(16))) char smem3[];
  int mt, nt;
  tile_of_block(blockIdx.x, nMt, nNt, mt, nt);
  w1p3_tile<EPI, F16, GATHER>(p, mt * BM, nt * BN, smem3);
}

// Workgroups [0, nMt * nNt) are whole tiles.  With S > 0 the grid continues with the K-split slices of the tail tiles
// (gemm_bf16x3_tail_plan: the M tiles nMt .. nMt + tail_mt - 1 of the last, nearly empty round): workgroup = (tile, slice),
// slice s accumulates channel blocks [s * ncb / S, (s + 1) * ncb / S) and stores its RAW accumulators to
// partial[s][row - nMt * 128][Npad]; bf16x3_tail_reduce_kernel adds the slices in order and runs the epilogue.  The slices
// are the highest workgroup ids, so they are dispatched into the slots the last whole tiles leave free -- as their own
// launch they ran latency-bound on an otherwise idle chip (17-21 us per layer for 1.4 % of its MFMA work).
template <int NPS, int EPI = 0, bool F16 = false>
__global__ __launch_bounds__(256, 3) void gemm_bf16x3_w14p2_kernel(GemmArgs p, int nMt, int nNt, int w, int S) {
  extern __shared__ __attribute__((aligned(16))) char smem3[];
  int mt, nt, cb_begin = 0, cb_end = (p.Kpad >> 5) / w;
  bool slice = false;
  if (EPI == 3 && (int)blockIdx.x >= nMt * nNt) {
    const int id = blockIdx.x - nMt * nNt;
    const int split = id % S, tile = id / S;
    mt = nMt + tile / nNt;
    nt = tile % nNt;
    const int per = cb_end / S;
    cb_begin = split * per;
    cb_end = cb_begin + per;
    p.Y = p.partial + ((int64_t)split * p.tail_mt - nMt) * (int64_t)BM * p.Npad;
    p.ldy = p.Npad;
    slice = true;
  } else if (EPI == 0 && (int)blockIdx.x >= nMt * nNt) {
    const int id = blockIdx.x - nMt * nNt;
    const int split = id % S, tile = id / S;
    mt = nMt + tile / nNt;
    nt = tile % nNt;
    const int per = cb_end / S;
    cb_begin = split * per;
    cb_end = cb_begin + per;
    p.Y = p.partial + ((int64_t)split * p.tail_mt - nMt) * (int64_t)BM * p.Npad;   // row m of the tile -> slice row m - nMt*128
    p.raw = 1;
    p.act = ACT_NONE;                    // raw partial sums: no scale / shift / activation before the reduce
    p.alpha = nullptr;
    p.ldy = p.Npad;
    p.N = p.Npad;
    p.M = (nMt + p.tail_mt) * BM;        // every row of the tail tiles is stored (rows >= M are never read back)
    p.Ysb = nullptr;
    p.rowmap = nullptr;
    p.R = nullptr;
    p.pool_part = nullptr;
  } else {
    tile_of_block(blockIdx.x, nMt, nNt, mt, nt);
  }
  w14p2_tile<NPS, EPI, F16>(p, mt * BM, nt * BN, w, smem3, cb_begin, cb_end, slice);
}

// one thread per (tail row, 4 channels): ordered sum of the K slices, then the usual epilogue (BN scale/shift,
// activation, rowmap compaction, fp32 and/or split-blocked store).  Padding channels (n >= N) are written as zeros.
__global__ void bf16x3_tail_reduce_kernel(GemmArgs p, int mt0, int S) {
  const int quads = p.Npad >> 2;
  const int64_t rows = (int64_t)p.tail_mt * BM;
  const int64_t total = rows * quads;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / quads;
    const int n = (int)(i - r * quads) * 4;
    const int64_t m = (int64_t)mt0 * BM + r;
    if (m >= p.M) continue;
    bool zero;
    const int orow = out_row(p, (int)m, zero);
    if (orow < 0) continue;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) acc += *reinterpret_cast<const f32x4*>(p.partial + ((int64_t)s * rows + r) * p.Npad + n);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (n + e < p.N && !zero) v[e] = apply_act(fmaf(acc[e], p.scale[n + e], p.shift[n + e]), p.act, p.alpha ? p.alpha[n + e] : 0.f);
    if (p.Y && n < p.N) *reinterpret_cast<f32x4*>(p.Y + (int64_t)orow * p.ldy + n) = v;     // N % 4 == 0 (wide epilogue)
    if (p.Ysb && n < p.ldsb) {
      uint32_t h01, l01, h23, l23;
      v *= p.sb_mul;
      split2(v[0], v[1], h01, l01, p.f16);
      split2(v[2], v[3], h23, l23, p.f16);
      if (p.f16) ovf_report(p.ovf, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
      char* blk = reinterpret_cast<char*>(p.Ysb) + (int64_t)orow * p.ldsb * 4 + (n >> 5) * 128 + (n & 31) * 2;
      *reinterpret_cast<uint2*>(blk) = make_uint2(h01, h23);
      *reinterpret_cast<uint2*>(blk + 64) = make_uint2(l01, l23);
    }
  }
}

#undef XV_GLD
#undef XV_WAIT2

#ifdef XV_GEMM_TRACE
namespace {
long long* g_trace = nullptr;      // debug trace buffer (device), kTraceWgs workgroups x 8 stamps (0-3: kernel phases, 4-7: inside the epilogue)
int g_trace_wgs = 0;
constexpr int kTraceWgs = 16384;
}  // namespace
#endif

int64_t gemm_bf16x3_tail_plan(int M, int Kpad, int Npad, int w, int* tail_mt, int* splits, int cb_quant) {
  *tail_mt = 0;
  *splits = 1;
  if (w < 1 || w > 9 || (Kpad >> 5) % w != 0) return 0;          // (switch: xv_set_option "tail_split")
  const int slots = 256 * 3;             // CUs x resident workgroups of the default kernel
  const int nMt = (M + BM - 1) / BM, nNt = Npad / BN, tiles = nMt * nNt, ncb = (Kpad >> 5) / w;
  const int r = tiles % slots;
  if (tiles <= slots || r == 0 || r > slots / 4) return 0;        // only a nearly empty last round is worth splitting
  const int mt = (r + nNt - 1) / nNt;
  int S = 0;
  for (int c = 8; c >= 2; c >>= 1)
    if (ncb % c == 0 && ncb / c >= 2 && (ncb / c) % cb_quant == 0 && mt * nNt * c <= 256) { S = c; break; }   // about one slice workgroup per CU
  if (S == 0) return 0;
  *tail_mt = mt;
  *splits = S;
  return (int64_t)S * mt * BM * Npad * (int64_t)sizeof(float);
}

hipError_t launch_gemm_bf16x3(const GemmArgs& a_in, hipStream_t s) {
  if (a_in.M <= 0) return hipSuccess;
  GemmArgs a = a_in;
#ifdef XV_GEMM_TRACE
  {
    static int trace_k = -1, trace_n = 0;
    if (trace_k < 0) {
      const char* e = getenv("XVEC_TRACE_K");
      trace_k = e ? atoi(e) : 0;
      const char* e2 = getenv("XVEC_TRACE_N");
      trace_n = e2 ? atoi(e2) : 0;
    }
    if (trace_k > 0 && a.K == trace_k && (trace_n == 0 || a.N == trace_n)) {
      if (!g_trace && hipMalloc(&g_trace, sizeof(long long) * 8 * kTraceWgs) != hipSuccess) g_trace = nullptr;
      const int wgs = ((a.M + BM - 1) / BM) * (a.Npad / BN);
      if (g_trace && wgs <= kTraceWgs) {
        a.trace = g_trace;
        g_trace_wgs = wgs;
        (void)hipMemsetAsync(g_trace, 0, sizeof(long long) * 8 * kTraceWgs, s);
      }
    }
  }
#endif
  static std::mutex init_mu;    // the attributes are per device; any thread may make the first launch on one
  static bool attr_set[64] = {};
  const size_t smemw32 = (size_t)2 * DA_BYTES;      // two slabs; the 4 x 8 KB epilogue scratch overlays them
  const size_t smemw1p3 = (size_t)3 * DA_BYTES;     // one-tap form: three slabs (52 KB, three workgroups per CU)
#ifdef XV_LAB
  static int force = 0;         // XVEC_GEMM_TILE (A/B): 128 register-staged | 1 LDS-DMA weights, barrier per step; 0 = default
  static int diag = 0;          // XVEC_GEMM_DIAG: timing-only ablation switches of the LDS-DMA kernel (outputs invalid)
  const size_t smem128 = (size_t)4 * TILE_B;
  const size_t smemdma = (size_t)2 * DA_BYTES + 2 * DB_BYTES;
#endif
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  {
    std::lock_guard<std::mutex> init_lock(init_mu);
    if (!attr_set[dev & 63]) {
      const void* kernels[] = {
          reinterpret_cast<const void*>(gemm_bf16x3_w14p2_kernel<1, 0, false>), reinterpret_cast<const void*>(gemm_bf16x3_w14p2_kernel<4, 0, false>),
          reinterpret_cast<const void*>(gemm_bf16x3_w14p2_kernel<4, 1, false>), reinterpret_cast<const void*>(gemm_bf16x3_w14p2_kernel<4, 2, false>),
          reinterpret_cast<const void*>(gemm_bf16x3_w14p2_kernel<1, 0, true>), reinterpret_cast<const void*>(gemm_bf16x3_w14p2_kernel<4, 0, true>),
          reinterpret_cast<const void*>(gemm_bf16x3_w14p2_kernel<4, 1, true>), reinterpret_cast<const void*>(gemm_bf16x3_w14p2_kernel<4, 2, true>),
          reinterpret_cast<const void*>(gemm_bf16x3_w14p2_kernel<1, 3, true>), reinterpret_cast<const void*>(gemm_bf16x3_w14p2_kernel<4, 3, true>)};
      hipError_t r = hipSuccess;
      for (const void* k : kernels) {
        r = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smemw32);
        if (r != hipSuccess) return r;
      }
      const void* kernels3[] = {
          reinterpret_cast<const void*>(gemm_bf16x3_w1p3_kernel<0, false>), reinterpret_cast<const void*>(gemm_bf16x3_w1p3_kernel<0, true>),
          reinterpret_cast<const void*>(gemm_bf16x3_w1p3_kernel<1, false>), reinterpret_cast<const void*>(gemm_bf16x3_w1p3_kernel<1, true>),
          reinterpret_cast<const void*>(gemm_bf16x3_w1p3_kernel<2, false>), reinterpret_cast<const void*>(gemm_bf16x3_w1p3_kernel<2, true>),
          reinterpret_cast<const void*>(gemm_bf16x3_w1p3_kernel<0, false, true>), reinterpret_cast<const void*>(gemm_bf16x3_w1p3_kernel<0, true, true>),
          reinterpret_cast<const void*>(gemm_bf16x3_w1p3_kernel<3, true>), reinterpret_cast<const void*>(gemm_bf16x3_w1p3_kernel<3, true, true>)};
      for (const void* k : kernels3) {
        r = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smemw1p3);
        if (r != hipSuccess) return r;
      }
#ifdef XV_LAB
      const char* e = getenv("XVEC_GEMM_TILE");
      force = e ? atoi(e) : 0;
      r = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem128);
      if (r != hipSuccess) return r;
      r = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_dma_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smemdma);
      if (r != hipSuccess) return r;
      const char* e3 = getenv("XVEC_GEMM_DIAG");
      diag = e3 ? atoi(e3) : 0;
#endif
      attr_set[dev & 63] = true;
    }
  }
  const int w = a.K / a.cin > 0 && a.ldsbx == a.cin ? a.K / a.cin : 1;   // taps (dense: 1)
  const bool taps_ok = w <= 9 && (a.Kpad >> 5) % w == 0;   // slab halo: 128 + w - 1 <= DA_ROWS (136)
  const int nNt = a.Npad / BN, nMt = (a.M + BM - 1) / BM;
#ifdef XV_LAB
  if (force == 1 && taps_ok) {
    hipLaunchKernelGGL(gemm_bf16x3_dma_kernel, dim3(nMt * nNt), dim3(256), smemdma, s, a, nMt, nNt, w, diag);
    return hipGetLastError();
  }
  if (force == 128 || !taps_ok) {
    hipLaunchKernelGGL(gemm_bf16x3_kernel, dim3(nMt * nNt), dim3(256), smem128, s, a, nMt, nNt);
    return hipGetLastError();
  }
#endif
  if (!taps_ok) return hipErrorInvalidValue;   // the plan only routes layers of <= 9 taps and whole 32-channel blocks here
  // 1 x 4 waves, weights in registers two steps ahead.  Tail handling is decided at plan time (the plan owns the
  // partial workspace): the last tail_mt M tiles go K-split
  const dim3 block(256);
  if (a.att_part || a.pool_w) {                 // attention epilogue: dense layers without row compaction only
    if (w != 1 || a.rowmap || a.a_pitch || a.R || (a.N & 3) || (a.pool_w && !a.pool_part)) return hipErrorInvalidValue;
    const dim3 grid(nMt * nNt);
    if (!a.slab3) {
      if (a.att_part) {
        if (a.f16) hipLaunchKernelGGL((gemm_bf16x3_w14p2_kernel<4, 1, true>), grid, block, smemw32, s, a, nMt, nNt, w, 0);
        else       hipLaunchKernelGGL((gemm_bf16x3_w14p2_kernel<4, 1, false>), grid, block, smemw32, s, a, nMt, nNt, w, 0);
      } else {
        if (a.f16) hipLaunchKernelGGL((gemm_bf16x3_w14p2_kernel<4, 2, true>), grid, block, smemw32, s, a, nMt, nNt, w, 0);
        else       hipLaunchKernelGGL((gemm_bf16x3_w14p2_kernel<4, 2, false>), grid, block, smemw32, s, a, nMt, nNt, w, 0);
      }
      return hipGetLastError();
    }
    if (a.att_part) {
      if (a.f16) hipLaunchKernelGGL((gemm_bf16x3_w1p3_kernel<1, true>), grid, block, smemw1p3, s, a, nMt, nNt);
      else       hipLaunchKernelGGL((gemm_bf16x3_w1p3_kernel<1, false>), grid, block, smemw1p3, s, a, nMt, nNt);
    } else {
      if (a.f16) hipLaunchKernelGGL((gemm_bf16x3_w1p3_kernel<2, true>), grid, block, smemw1p3, s, a, nMt, nNt);
      else       hipLaunchKernelGGL((gemm_bf16x3_w1p3_kernel<2, false>), grid, block, smemw1p3, s, a, nMt, nNt);
    }
    return hipGetLastError();
  }
  const bool tail = a.tail_mt > 0 && a.ksplit > 1 && a.partial && !a.a_pitch && !a.pool_part && !a.R && !a.raw &&
                    a.tail_mt < nMt && (a.N & 3) == 0;
  const int nMain = tail ? nMt - a.tail_mt : nMt;
  const dim3 grid(nMain * nNt);
#ifdef XV_GEMM_TRACE
  if (a.trace) g_trace_wgs = (int)grid.x <= kTraceWgs ? (int)grid.x : 0;
#endif
  if (a.arow) {                                 // gathered grid rows (ResNet convolutions without border rows)
    if (w != 1 || tail || !a.a_pitch || a.pool_part) return hipErrorInvalidValue;
    if (a.ysb_f6) {                             // the reader is a two-unit ResNet convolution: its block format, nothing else
      if (!a.f16 || !a.Ysb || a.Y || a.R) return hipErrorInvalidValue;
      hipLaunchKernelGGL((gemm_bf16x3_w1p3_kernel<3, true, true>), grid, block, smemw1p3, s, a, nMain, nNt);
      return hipGetLastError();
    }
    if (a.f16) hipLaunchKernelGGL((gemm_bf16x3_w1p3_kernel<0, true, true>), grid, block, smemw1p3, s, a, nMain, nNt);
    else       hipLaunchKernelGGL((gemm_bf16x3_w1p3_kernel<0, false, true>), grid, block, smemw1p3, s, a, nMain, nNt);
    return hipGetLastError();
  }
  if (a.ysb_f6 && (!a.f16 || (w > 1 && w < 5) || !a.Ysb || a.Y || a.R || a.pool_part)) return hipErrorInvalidValue;
  if (w == 1 && !tail && a.slab3) {             // one tap: three slab buffers, slabs two steps ahead
    if (a.ysb_f6) hipLaunchKernelGGL((gemm_bf16x3_w1p3_kernel<3, true>), grid, block, smemw1p3, s, a, nMain, nNt);
    else if (a.f16) hipLaunchKernelGGL((gemm_bf16x3_w1p3_kernel<0, true>), grid, block, smemw1p3, s, a, nMain, nNt);
    else       hipLaunchKernelGGL((gemm_bf16x3_w1p3_kernel<0, false>), grid, block, smemw1p3, s, a, nMain, nNt);
    return hipGetLastError();
  }
  // whole tiles, then (tail form) the K-split slices of the tail tiles in the same launch
  const int S = tail ? a.ksplit : 0;
  const dim3 grid2(nMain * nNt + (tail ? a.tail_mt * nNt * a.ksplit : 0));
  if (a.ysb_f6) {                               // XV_PREC_F16F6: this layer's only reader is a two-unit layer -> its block format
    // (whole tiles through the block-format epilogue, the K-split slices as raw sums that the two-unit kernel's reduce finishes)
    if (w >= 5) hipLaunchKernelGGL((gemm_bf16x3_w14p2_kernel<1, 3, true>), grid2, block, smemw32, s, a, nMain, nNt, w, S);
    else        hipLaunchKernelGGL((gemm_bf16x3_w14p2_kernel<4, 3, true>), grid2, block, smemw32, s, a, nMain, nNt, w, S);
    const hipError_t e6 = hipGetLastError();
    if (e6 != hipSuccess || !tail) return e6;
    return launch_f6v2_tail_reduce(a, nMain, s);
  }
  if (w >= 5) {
    if (a.f16) hipLaunchKernelGGL((gemm_bf16x3_w14p2_kernel<1, 0, true>), grid2, block, smemw32, s, a, nMain, nNt, w, S);
    else       hipLaunchKernelGGL((gemm_bf16x3_w14p2_kernel<1, 0, false>), grid2, block, smemw32, s, a, nMain, nNt, w, S);
  } else {
    if (a.f16) hipLaunchKernelGGL((gemm_bf16x3_w14p2_kernel<4, 0, true>), grid2, block, smemw32, s, a, nMain, nNt, w, S);
    else       hipLaunchKernelGGL((gemm_bf16x3_w14p2_kernel<4, 0, false>), grid2, block, smemw32, s, a, nMain, nNt, w, S);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || !tail) return e;
  const int64_t total = (int64_t)a.tail_mt * BM * (a.Npad >> 2);
  hipLaunchKernelGGL(bf16x3_tail_reduce_kernel, dim3((unsigned)((total + 255) / 256)), block, 0, s, a, nMain, a.ksplit);
  return hipGetLastError();
}

}  // namespace xv

#ifdef XV_GEMM_TRACE
// debug only (not part of the public ABI): copy the phase stamps of the last traced GEMM launch
extern "C" int xvdbg_gemm_trace(long long* out, int max_wgs) {
  if (!xv::g_trace || xv::g_trace_wgs <= 0) return 0;
  const int n = xv::g_trace_wgs < max_wgs ? xv::g_trace_wgs : max_wgs;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpy(out, xv::g_trace, sizeof(long long) * 8 * n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return n;
}
#endif
