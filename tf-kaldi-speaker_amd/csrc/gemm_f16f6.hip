// Two-unit split-precision GEMM for the multi-tap temporal convolutions (XV_PREC_F16F6), gfx950.
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
//   chunk 4 / 5 bytes 0-15 of the 32 x 6-bit codes of q6(hi) / q6(lo)      chunk 6  their bytes 16-23 (hi | lo)
//   chunk 7     byte 0 / 1 of EVERY dword: the E8M0 scales of q6(hi) / q6(lo) (four copies, so that the reader can take the dword
//               (row & 1) | ((row >> 4) & 1) << 1 and its 4-byte reads of 17 consecutive rows fall on 17 different banks)
// One lane's operand of the scaled MFMA is one such block: lane l holds row / column l & 15 and K group l >> 4 = 32 consecutive K
// elements, 32 x 6 bits little-endian, scale per lane (checked bit-exactly: tools/mfma_scale_probe.hip).  The K = 128 of a scaled
// MFMA is "four taps of one channel block": group g reads slab row frame + 4q + g -- the rows the four main MFMAs of those taps read.
// A channel block is two macro steps [cross pass: 32 scaled MFMAs | main pass: 16 MFMAs per real tap]; taps 7 (and 5, 6 of a
// 5-tap layer) have zero weights in the cross operands and are skipped in the main pass.
//
// Schedule (the prototype's, tools/proto/f6_gemm.hip): slab staging by LDS-DMA exactly as gemm_bf16x3_w14p2_kernel (two buffers,
// XOR-swizzled 16-byte chunks); main weights double-buffered in registers one macro step ahead; the cross weights of the next macro
// step are loaded behind the cross pass into the registers it has just finished with; one s_waitcnt vmcnt(0) per macro step (the
// loads were issued a whole pass earlier); 246 VGPRs, two workgroups per CU.
#include <mutex>

#include "xv_f6.h"

namespace xv {

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lptr6_t;

constexpr int F6_BM = 128, F6_BN = 128, F6_DROW = 128, F6_DA_ROWS = 136, F6_DA_BYTES = F6_DA_ROWS * F6_DROW;
constexpr int64_t kWmainCt = 64 * 16, kWxCt = 64 * 16 * 3 + 64 * 4;

}  // namespace

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
  dst[6] = uint4{ch[4], ch[5], cl[4], cl[5]};
  const uint32_t sc2 = bh | (bl << 8);
  dst[7] = uint4{sc2, sc2, sc2, sc2};        // four copies: the reader picks a dword by row (LDS bank spread)
}

#define XV6_GLD16(dst, ptr, OFF) asm volatile("global_load_dwordx4 %0, %1, off offset:" #OFF : "=v"(dst) : "v"(ptr))
#define XV6_GLD8(dst, ptr, OFF) asm volatile("global_load_dwordx2 %0, %1, off offset:" #OFF : "=v"(dst) : "v"(ptr))
#define XV6_GLD4(dst, ptr, OFF) asm volatile("global_load_dword %0, %1, off offset:" #OFF : "=v"(dst) : "v"(ptr))

// NTAPS = real taps of the layer (4 .. 8); the weights are packed for 8 (tap >= NTAPS: zeros)
template <int NTAPS>
__global__ __launch_bounds__(256, 2) void gemm_f16f6_kernel(GemmArgs p, int nMt, int nNt) {
  extern __shared__ __attribute__((aligned(16))) char smem6[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g4 = lane >> 4;
  const int tile = xcd_remap(blockIdx.x, nMt * nNt);
  const int mt = tile / nNt, nt = tile - mt * nNt;
  const int m0 = mt * F6_BM, n0 = nt * F6_BN;
  const int lrow = lane >> 3, lpc = lane & 7;
  const int ncb = p.cin >> 5;                                   // channel blocks per frame
  const int64_t a_row_bytes = p.ldsbx * 4;
  const char* Ag = reinterpret_cast<const char*>(p.Xsb) + (int64_t)(m0 + lrow) * a_row_bytes;
  const uint32_t as_lds = (uint32_t)(uintptr_t)(lptr6_t)smem6;
  // Slab swizzle (conflict-free for every tap shift and for the cross reads: tests/analysis/lds_bank_model.py).  With p = r >> 1:
  //   chunks 0-3 (f16 hi)    at  c ^ f1(r),  f1 = 2 * (p & 3)                      -- a ds_read_b128 group is 16 consecutive rows of which
  //                           the outer eight read k chunk g4 and the inner eight g4 ^ 1: bit 0 of the chunk must survive the swizzle;
  //   chunks 4-7 (fp6, scale) at c ^ f2(r),  f2 = (p1 << 2) | (p2 << 1) | p0         -- all lanes read the SAME chunk of 16 consecutive
  //                           rows: eight rows of one parity need eight positions.
  // Bit 2 of f1 and f2 agree (p1), so the two halves of a row stay disjoint.  Row r = 8 g + lrow: p & 3 = lrow >> 1, p2 = g & 1.
  const int swz1 = ((lrow >> 1) & 3) << 1;
  const int swz2 = (((lrow >> 2) & 1) << 2) | ((lrow >> 1) & 1);                 // f2 without its bit 1 (= g & 1)
  auto dma_a = [&](int cb, int buf, int g) {                    // 8 rows x 128 B of channel block cb
    // LDS position lpc of slab row r = 8 g + lrow holds chunk  lpc ^ f1(r)  or  lpc ^ f2(r)  (slab swizzle, below)
    const int c = lpc ^ ((((lpc >> 2) ^ (lrow >> 2)) & 1) ? (swz2 ^ ((g & 1) << 1)) : swz1);
    const char* src = Ag + (int64_t)(8 * g) * a_row_bytes + (int64_t)cb * 128 + c * 16;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(as_lds + buf * F6_DA_BYTES + g * 1024);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory");
  };
  const char* Wm_g = reinterpret_cast<const char*>(p.Wfr) + ((int64_t)((n0 >> 5) + wave) * ncb * 8 * 2) * kWmainCt + lane * 16;
  const char* Wx_g = reinterpret_cast<const char*>(p.Wx6) + ((int64_t)((n0 >> 5) + wave) * ncb * 2 * 2) * kWxCt;

  f32x4 acc[8][2];
#pragma unroll
  for (int g = 0; g < 8; ++g)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[g][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  f16x8 Wm[2][4][2];          // main weights [buffer][tap of the macro step][channel tile], one macro step ahead
  v4i Xh[2], Xl[2];           // cross weights of the CURRENT macro step: q6(hi) / q6(lo) bytes 0-15 per channel tile; reloaded for
  v2i Xth[2], Xtl[2];         // the next step right behind the cross pass, while the main pass runs
  int Xs[2];                  // scale bytes: 0 = q6(hi), 1 = q6(lo)
  // Only the real taps are loaded: the destination of a load nobody reads is a dead register to the compiler, which hands it
  // to something else while the load is still in flight (seen as an intermittent fault on a corrupted address register).
  // buf == macro step type q at every call site (step even <-> q = 0 <-> buffer 0), a literal after inlining.
  auto load_wm = [&](int step, int buf) __attribute__((always_inline)) {      // step = cb * 2 + q; 8 KB per wave and step
    constexpr int NT0 = NTAPS < 4 ? NTAPS : 4;
    const int nt = buf == 0 ? NT0 : NTAPS - 4;
    const char* pm = Wm_g + (int64_t)step * (4 * 2) * kWmainCt;
    if (0 < nt) { XV6_GLD16(Wm[buf][0][0], pm, 0);    XV6_GLD16(Wm[buf][0][1], pm, 1024); }
    if (1 < nt) { XV6_GLD16(Wm[buf][1][0], pm, 2048); XV6_GLD16(Wm[buf][1][1], pm, 3072); }
    const char* pm2 = pm + 4096;
    if (2 < nt) { XV6_GLD16(Wm[buf][2][0], pm2, 0);    XV6_GLD16(Wm[buf][2][1], pm2, 1024); }
    if (3 < nt) { XV6_GLD16(Wm[buf][3][0], pm2, 2048); XV6_GLD16(Wm[buf][3][1], pm2, 3072); }
  };
  // every register an in-flight load writes is an operand of the wait, so the value is not "available" to the compiler before it
  auto wait_all = [&](int buf) __attribute__((always_inline)) {
    constexpr int NT0 = NTAPS < 4 ? NTAPS : 4;
    const int nt = buf == 0 ? NT0 : NTAPS - 4;
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(Xh[0]), "+v"(Xh[1]), "+v"(Xl[0]), "+v"(Xl[1]), "+v"(Xth[0]), "+v"(Xth[1]), "+v"(Xtl[0]), "+v"(Xtl[1]),
                   "+v"(Xs[0]), "+v"(Xs[1])
                 :
                 : "memory");
    if (0 < nt) asm volatile("" : "+v"(Wm[buf][0][0]), "+v"(Wm[buf][0][1]));
    if (1 < nt) asm volatile("" : "+v"(Wm[buf][1][0]), "+v"(Wm[buf][1][1]));
    if (2 < nt) asm volatile("" : "+v"(Wm[buf][2][0]), "+v"(Wm[buf][2][1]));
    if (3 < nt) asm volatile("" : "+v"(Wm[buf][3][0]), "+v"(Wm[buf][3][1]));
  };
  // the same, leaving the five youngest operations (the slab pieces issued behind the cross weights) in flight
  auto wait_keep5 = [&](int buf) __attribute__((always_inline)) {
    constexpr int NT0 = NTAPS < 4 ? NTAPS : 4;
    const int nt = buf == 0 ? NT0 : NTAPS - 4;
    asm volatile("s_waitcnt vmcnt(5)"
                 : "+v"(Xh[0]), "+v"(Xh[1]), "+v"(Xl[0]), "+v"(Xl[1]), "+v"(Xth[0]), "+v"(Xth[1]), "+v"(Xtl[0]), "+v"(Xtl[1]),
                   "+v"(Xs[0]), "+v"(Xs[1])
                 :
                 : "memory");
    if (0 < nt) asm volatile("" : "+v"(Wm[buf][0][0]), "+v"(Wm[buf][0][1]));
    if (1 < nt) asm volatile("" : "+v"(Wm[buf][1][0]), "+v"(Wm[buf][1][1]));
    if (2 < nt) asm volatile("" : "+v"(Wm[buf][2][0]), "+v"(Wm[buf][2][1]));
    if (3 < nt) asm volatile("" : "+v"(Wm[buf][3][0]), "+v"(Wm[buf][3][1]));
  };
  auto load_x = [&](int step) __attribute__((always_inline)) {
    // per channel tile 3 328 bytes: [64 x 16 hi | 64 x 16 lo | 64 x (8 hi tail | 8 lo tail) | 64 x 4 scales]
    const char* p0 = Wx_g + (int64_t)step * 2 * kWxCt + lane * 16;
    XV6_GLD16(Xh[0], p0, 0); XV6_GLD16(Xl[0], p0, 1024); XV6_GLD8(Xth[0], p0, 2048); XV6_GLD8(Xtl[0], p0, 2056);
    const char* p1 = p0 + kWxCt;
    XV6_GLD16(Xh[1], p1, 0); XV6_GLD16(Xl[1], p1, 1024); XV6_GLD8(Xth[1], p1, 2048); XV6_GLD8(Xtl[1], p1, 2056);
    const char* ps = Wx_g + (int64_t)step * 2 * kWxCt + 3072 + lane * 4;
    XV6_GLD4(Xs[0], ps, 0);
    const char* ps1 = ps + kWxCt;
    XV6_GLD4(Xs[1], ps1, 0);
  };
  const int nsteps = ncb * 2;
  // prologue: slab 0, weights of step 0
  for (int g = wave; g < 17; g += 4) dma_a(0, 0, g);
  load_wm(0, 0);
  load_x(0);
  wait_all(0);
  __syncthreads();

  // LDS offsets of this lane's fragments inside a slab, per macro step type q (the swizzle (row >> 1) & 7 does not depend on
  // the frame tile: 16 rows further is + 2048 bytes, an immediate)
  int xo[2][4], mo[2][4];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int rx = c16 + 4 * q + g4, px = rx >> 1;             // cross: K group g4 = tap 4q + g4
    const int sx = (((px >> 1) & 1) << 2) | (((px >> 2) & 1) << 1) | (px & 1);     // f2(rx); 16 rows further: the same
    xo[q][0] = rx * F6_DROW + ((4 ^ sx) << 4);
    xo[q][1] = rx * F6_DROW + ((5 ^ sx) << 4);
    xo[q][2] = rx * F6_DROW + ((6 ^ sx) << 4);
    // scale dword (rx & 1) | ((row >> 4) & 1) << 1 of chunk 7, row = 16 tile + rx: for odd tiles flip address bit 3
    xo[q][3] = rx * F6_DROW + ((7 ^ sx) << 4) + (((rx & 1) | (((rx >> 4) & 1) << 1)) << 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = c16 + 4 * q + j;
      mo[q][j] = r * F6_DROW + ((g4 ^ (((r >> 1) & 3) << 1)) << 4);
    }
  }
  // cross pass: the two block-scaled MFMAs of every tile.  Four MFMAs (64 cycles) per tile do not cover an LDS read, so the
  // fragments are read XD - 1 tiles ahead: two where the registers allow it (up to 6 taps), one otherwise
  constexpr int XD = NTAPS <= 6 ? 4 : 3;
  auto cross_pass = [&](int cb, int q) __attribute__((always_inline)) {
    const char* slab = smem6 + (cb & 1) * F6_DA_BYTES;
    v4i fh[XD], fl[XD], ft[XD];              // codes 0-15 of q6(hi) / q6(lo); tails (hi | lo): one ds_read_b128 each, conflict-free
    int fs[XD];
    auto read_cross = [&](int g, int slot) __attribute__((always_inline)) {
      fh[slot] = *reinterpret_cast<const v4i*>(slab + xo[q][0] + g * 2048);
      fl[slot] = *reinterpret_cast<const v4i*>(slab + xo[q][1] + g * 2048);
      ft[slot] = *reinterpret_cast<const v4i*>(slab + xo[q][2] + g * 2048);
      fs[slot] = *reinterpret_cast<const int*>(slab + (xo[q][3] ^ ((g & 1) << 3)) + g * 2048);
    };
#pragma unroll
    for (int g = 0; g < XD - 1; ++g) read_cross(g, g);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int sl = g % XD;
      if (g + XD - 1 < 8) read_cross(g + XD - 1, (g + XD - 1) % XD);
      const v8i a_hi6 = {fh[sl][0], fh[sl][1], fh[sl][2], fh[sl][3], ft[sl][0], ft[sl][1], 0, 0};
      const v8i a_lo6 = {fl[sl][0], fl[sl][1], fl[sl][2], fl[sl][3], ft[sl][2], ft[sl][3], 0, 0};
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const v8i w_hi6 = {Xh[c][0], Xh[c][1], Xh[c][2], Xh[c][3], Xth[c][0], Xth[c][1], 0, 0};
        const v8i w_lo6 = {Xl[c][0], Xl[c][1], Xl[c][2], Xl[c][3], Xtl[c][0], Xtl[c][1], 0, 0};
        // w_lo * a_hi (weight scale byte 1, activation scale byte 0), then w_hi * a_lo (bytes 0 / 1)
        acc[g][c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w_lo6, a_hi6, acc[g][c], 2, 2, 1, Xs[c], 0, fs[sl]);
        acc[g][c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w_hi6, a_lo6, acc[g][c], 2, 2, 0, Xs[c], 1, fs[sl]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // main pass: hi * hi of the real taps of the macro step
  auto main_pass = [&](int cb, int q, int buf) __attribute__((always_inline)) {
    const char* slab = smem6 + (cb & 1) * F6_DA_BYTES;
    constexpr int NT0 = NTAPS < 4 ? NTAPS : 4;
    const int nt_here = q == 0 ? NT0 : NTAPS - 4;          // compile-time after inlining (q is a literal at both call sites)
    constexpr int MD = NTAPS <= 7 ? 3 : 2;               // fragment slots: read MD - 1 tiles ahead
    f16x8 fm[MD][4];
    auto read_main = [&](int g, int slot) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < nt_here) fm[slot][j] = *reinterpret_cast<const f16x8*>(slab + mo[q][j] + g * 2048);
    };
#pragma unroll
    for (int g = 0; g < MD - 1; ++g) read_main(g, g);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int sl = g % MD;
      if (g + MD - 1 < 8) read_main(g + MD - 1, (g + MD - 1) % MD);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < nt_here) {
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wm[buf][j][c], fm[sl][j], acc[g][c], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // the whole next slab goes out in the FIRST macro step of a channel block, behind the cross weights: five pieces per wave (17
  // groups over 4 waves, the last ones clamped duplicates), so that the wait at the end of that step can leave exactly them in
  // flight -- they are only needed at the barrier one macro step later
  auto dma_next = [&](int cb) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int g = wave + 4 * i;
      dma_a(cb + 1 < ncb ? cb + 1 : cb, (cb + 1) & 1, g < 17 ? g : 16);
    }
  };
  for (int step = 0; step < nsteps; step += 2) {        // two macro steps = one channel block; main weight buffers alternate statically
    const int cb = step >> 1;
    // ---- q = 0: main weights of (cb, 1) first; behind the cross pass the cross weights of (cb, 1), then slab cb + 1
#ifndef XV_F6_NOW      // (timing lab: -DXV_F6_NOW never reloads the main weights -- wrong results)
    load_wm(step + 1, 1);
#endif
    __builtin_amdgcn_sched_barrier(0);
    cross_pass(cb, 0);
    __builtin_amdgcn_sched_barrier(0);
#ifndef XV_F6_NOX      // (timing lab: -DXV_F6_NOX keeps the prologue's cross weights for every step -- wrong results)
    load_x(step + 1);                                   // land while the main pass runs
#endif
#ifndef XV_F6_NOD      // (timing lab: -DXV_F6_NOD stages no slab in the loop -- wrong results)
    dma_next(cb);
#endif
    __builtin_amdgcn_sched_barrier(0);
    main_pass(cb, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#ifndef XV_F6_NOD
    wait_keep5(1);                                      // weights landed; the five slab pieces may still be in flight
#else
    wait_all(1);
#endif
    // ---- q = 1
#ifndef XV_F6_NOW
    load_wm(step + 2 < nsteps ? step + 2 : 0, 0);
#endif
    __builtin_amdgcn_sched_barrier(0);
    cross_pass(cb, 1);
    __builtin_amdgcn_sched_barrier(0);
#ifndef XV_F6_NOX
    load_x(step + 2 < nsteps ? step + 2 : 0);
#endif
    __builtin_amdgcn_sched_barrier(0);
    main_pass(cb, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    wait_all(0);
    __syncthreads();                                    // slab cb + 1 complete; every wave is done reading slab cb
  }
  // (the last barrier of the loop is the one in front of the epilogue: no slab piece is in flight, every wave is done reading)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  if (p.ysb_f6 && p.Ysb && !p.Y) store_wave_tile_n32_f6(p, acc, m0, n0 + wave * 32, lane_e, wave, smem6);
  else store_wave_tile_n32<64, true>(p, acc, m0, n0 + wave * 32, lane_e, wave, smem6);
}

hipError_t launch_f6_from_sb(const void* sb, void* out, int64_t rows, int nblk, hipStream_t s) {
  const int64_t total = rows * nblk;
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(f6_from_sb_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, static_cast<const char*>(sb),
                     static_cast<char*>(out), rows, nblk);
  return hipGetLastError();
}

// a.Xsb = activations in the block format above (row stride a.ldsbx channels), a.Wfr / a.Wx6 = main / cross weights
// (xvec_api.hip, upload_layer), a.K = taps * a.cin, taps 5 or 7, a.cin % 32 == 0
hipError_t launch_gemm_f16f6(const GemmArgs& a, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  const int taps = a.cin > 0 ? a.K / a.cin : 0;
  // Only the widths the reference's graphs contain are instantiated (tdnn: 5, 5, 7; extended tdnn: 5, 5, 7 and a 9-tap layer that
  // stays on the three-unit kernel): an 8-tap instantiation sits at 256 VGPRs with no headroom for its hand-counted waits.
  if ((taps != 5 && taps != 7) || (a.cin & 31) || a.ldsbx != a.cin || !a.Wx6 || !a.Wfr || a.a_pitch || a.pool_part || a.R || (a.N & 3))
    return hipErrorInvalidValue;
  static std::mutex mu;
  static bool attr_set[64] = {};
  const size_t smem = (size_t)2 * F6_DA_BYTES;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (!attr_set[dev & 63]) {
      const void* ks[] = {reinterpret_cast<const void*>(gemm_f16f6_kernel<5>), reinterpret_cast<const void*>(gemm_f16f6_kernel<7>)};
      for (const void* k : ks) {
        const hipError_t r = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (r != hipSuccess) return r;
      }
      attr_set[dev & 63] = true;
    }
  }
  const int nMt = (a.M + F6_BM - 1) / F6_BM, nNt = a.Npad / F6_BN;
  const dim3 grid(nMt * nNt), block(256);
  if (taps == 5) hipLaunchKernelGGL(gemm_f16f6_kernel<5>, grid, block, smem, s, a, nMt, nNt);
  else hipLaunchKernelGGL(gemm_f16f6_kernel<7>, grid, block, smem, s, a, nMt, nNt);
  return hipGetLastError();
}

}  // namespace xv
