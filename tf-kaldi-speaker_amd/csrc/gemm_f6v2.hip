// Two-unit split-precision GEMM for the 5- and 7-tap temporal convolutions (XV_PREC_F16F6), second schedule: three workgroups
// per CU, every load two phases ahead of its use, counted waits only.  The arithmetic is that of gemm_f16f6.hip:
//
//   a * w ~ f16(a) * f16(w)                                                   v_mfma_f32_16x16x32_f16
//         + q6(f16(a)) * q6(w - f16(w)) + q6(a - f16(a)) * q6(f16(w))          2 x v_mfma_scale_f32_16x16x128_f8f6f4 (e2m3, E8M0 block scale)
//
// and so are the activation block format (gemm_f16f6.hip; chunks 6 / 7 = [8 B code tail | scale dword | pad] of q6(hi) / q6(lo)),
// the slab staging by LDS-DMA and the 128 x 128 workgroup tile of four waves, each wave one 32-channel block x 128 frames.
//
// What is different -- and why.  The first schedule held the main weights of two macro steps (64 registers) plus the cross weights
// (26) and ran at 250 VGPRs = two workgroups per CU, with one full vmcnt(0) drain per macro step; the counters said the matrix pipe
// was busy 54 % of the time, the LDS 36 % (bank conflicts removed: no change -- it was never the LDS), i.e. two waves per SIMD do
// not cover the waits, and 2 336 tiles on 512 slots are 4.56 rounds.  Here a channel block is EIGHT PHASES, each with its own
// small weight set, loaded while the two phases before it run into registers the phase before those has just released:
//
//   7 taps   XA0  XB0  M01  M23  XA1  XB1  M45  M6          XA / XB: the cross terms of a macro step q (four taps) as two
//   5 taps   XA0  XB0  M01  M23  XA1  XB1  M4                        sub-phases -- w_lo6 x a_hi6, then w_hi6 x a_lo6 -- of 16 scaled MFMAs:
//                                                                    one 16-register weight set and one 8-register fragment each;
//                                                           Mjk:     hi * hi of taps j, k: 16 weight registers, 32 MFMAs.
//
// Register sets XA, XB, WA, WB of 16 registers each = 64 instead of 90, which with the
// 64 accumulators and the fragments fits the 168 registers of three workgroups per CU (768 slots: 3.04 rounds, and the K-split tail
// of gemm_bf16x3.hip for the rest).  Every set is issued at the START of the phase after its last reader and consumed two or three
// phases later (48 - 80 MFMAs = 0.8 - 1.3 k cycles of this wave, three times that in wall time with three waves per SIMD); the waits
// are exact vmcnt(N) counts of the younger operations (table at the loop), never a drain.  The slab of the next channel block (five
// LDS-DMA pieces per wave) goes out right behind the workgroup barrier and has the whole channel block to land.
#include <mutex>
#include <type_traits>

#include "xv_f6.h"

namespace xv {

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr2_t;

constexpr int V2_BM = 128, V2_BN = 128, V2_DROW = 128;
// slab rows: 128 frames + the halo of the last tap group (a group of four taps reads up to row 15 + 4 q + 3 of the last tile), in
// whole 8-row DMA pieces: 136 rows for two macro steps (5 / 7 taps), 144 for three (9 taps)
constexpr int v2_groups(int ntaps) { return ntaps <= 8 ? 17 : 18; }
constexpr int v2_slab_bytes(int ntaps) { return v2_groups(ntaps) * 8 * V2_DROW; }
constexpr int64_t kMainCt = 64 * 16;               // bytes per (tap, 16-channel tile) of the main weights: 64 lanes x 16 B
constexpr int64_t kXCt = 2 * 64 * 16;              // per (macro step, term, 16-channel tile) of the cross weights: 64 x 16 B codes 0-15 |
                                                   // 64 x 16 B {codes 16-23, scale dword, pad}: one lane offset serves both planes

struct XSet { v4i c[2]; v4i t[2]; };               // per channel tile: bytes 0-15 of the 6-bit codes; {bytes 16-23, E8M0 scale (byte 0), pad}:
                                                   // two halves of one 8-register operand tuple (the scaled MFMA reads the first six)
struct WSet { f16x8 w[2][2]; };                    // [tap of the pair][channel tile]

}  // namespace

#define V2_GLD16(dst, voff, sbase, OFF) asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #OFF : "=v"(dst) : "v"(voff), "s"(sbase))
// the waits name the registers they release, so that no MFMA of the phase can be scheduled above them
#define V2_WAITX(N, X) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(X.c[0]), "+v"(X.c[1]), "+v"(X.t[0]), "+v"(X.t[1]))
#define V2_WAITW2(N, W) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(W.w[0][0]), "+v"(W.w[0][1]), "+v"(W.w[1][0]), "+v"(W.w[1][1]))
#define V2_WAITW1(N, W) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(W.w[0][0]), "+v"(W.w[0][1]))

template <int NTAPS, bool OUT_F6>
__device__ __forceinline__ void f6v2_tile(const GemmArgs& p, int m0, int n0, char* smem, int cb_begin, int cb_end, bool slice) {
  static_assert(NTAPS == 5 || NTAPS == 7 || NTAPS == 9, "taps");
  constexpr int NQ = (NTAPS + 3) / 4;    // macro steps (groups of four taps) per channel block
  constexpr int NGRP = v2_groups(NTAPS), V2_DA_BYTES = v2_slab_bytes(NTAPS);
  constexpr int NSLOT = 2;               // fragment slots per phase: tile g + 1 is read while tile g multiplies
  const int tid = threadIdx.x;
  int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ncb = p.cin >> 5;                                   // channel blocks per frame
  const int64_t a_row_bytes = p.ldsbx * 4;
  const uint32_t as_lds = (uint32_t)(uintptr_t)(lptr2_t)smem;
  // Slab swizzle (conflict-free for every tap shift and for the cross reads: tests/analysis/lds_bank_model.py).  With p = r >> 1:
  //   chunks 0-3 (f16 hi)     at  c ^ f1(r),  f1 = 2 * (p & 3)                 -- a ds_read_b128 group is 16 consecutive rows of which the
  //                            outer eight read k chunk g4 and the inner eight g4 ^ 1: bit 0 of the chunk must survive the swizzle;
  //   chunks 4-7 (fp6, scale) at  c ^ f2(r),  f2 = (p1 << 2) | (p2 << 1) | p0    -- all lanes read the SAME chunk of 16 consecutive rows:
  //                            eight rows of one parity need eight positions.
  // Bit 2 of f1 and f2 agree (p1), so the two halves of a row stay disjoint.  Row r = 8 g + lrow: p & 3 = lrow >> 1, p2 = g & 1.
  // A DMA piece = uniform base (scalar: tile row 8 g, channel block) + this lane's offset (its row of the eight, the chunk its LDS
  // position holds); only bit 1 of f2 depends on the piece, so the offset of an odd piece is the even one with chunk bit 1 flipped
  // where the position holds a chunk of the upper half.
  const char* Abase = reinterpret_cast<const char*>(p.Xsb) + (int64_t)m0 * a_row_bytes;
  // The loop keeps ONE lane-dependent register (the lane id): every per-lane offset below is recomputed from it where it is used
  // (a handful of VALU operations per phase, which the matrix pipe hides) instead of living in ~15 loop-invariant registers --
  // the difference between fitting 168 registers and spilling.  The empty asm makes the value opaque, so the compiler cannot
  // hoist what is derived from it.
  auto lane_now = [&]() __attribute__((always_inline)) {
    asm volatile("" : "+v"(lane));        // (in place: no copy of the register)
    return lane;
  };
  auto dma_a = [&](int cb, int buf, int g, int l) {             // 8 rows x 128 B of channel block cb; l = lane_now()
    const int lr = l >> 3, lp = l & 7;
    const int s1 = ((lr >> 1) & 3) << 1;
    const int s2 = (((lr >> 2) & 1) << 2) | ((lr >> 1) & 1) | ((g & 1) << 1);
    const int c = lp ^ ((((lp >> 2) ^ (lr >> 2)) & 1) ? s2 : s1);
    const uint32_t off = (uint32_t)lr * (uint32_t)a_row_bytes + (uint32_t)(c << 4);
    const char* base = Abase + (int64_t)(8 * g) * a_row_bytes + (int64_t)cb * 128;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(as_lds + buf * V2_DA_BYTES + g * 1024);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(dst) : "memory");
  };
  auto dma_next = [&](int cb) __attribute__((always_inline)) {  // slab cb + 1: five pieces per wave (17 groups, the last ones duplicates)
    const int nx = cb + 1;
    const int l = lane_now();
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int g = wave + 4 * i;
      dma_a(nx, (cb + 1) & 1, g < NGRP ? g : NGRP - 1, l);
    }
  };
  // weights of this wave's 32-channel block: uniform bases (scalar registers) + one lane offset each
  const int nb = (n0 >> 5) + wave;
  const char* Wm = reinterpret_cast<const char*>(p.Wfr) + (int64_t)nb * ncb * (8 * NQ) * kMainCt;     // [cb][4 NQ taps][2 tiles][1 KB]
  const char* Wx = reinterpret_cast<const char*>(p.Wx6) + (int64_t)nb * ncb * (4 * NQ) * kXCt;        // [cb][q][term][2 tiles][2 KB]
  auto load_w2 = [&](WSet& W, int cb, int tap0) __attribute__((always_inline)) {
    const int voA = lane_now() << 4;
    const char* b = Wm + ((int64_t)cb * (8 * NQ) + tap0 * 2) * kMainCt;
    V2_GLD16(W.w[0][0], voA, b, 0); V2_GLD16(W.w[0][1], voA, b, 1024);
    V2_GLD16(W.w[1][0], voA, b, 2048); V2_GLD16(W.w[1][1], voA, b, 3072);
  };
  auto load_w1 = [&](WSet& W, int cb, int tap0) __attribute__((always_inline)) {
    const int voA = lane_now() << 4;
    const char* b = Wm + ((int64_t)cb * (8 * NQ) + tap0 * 2) * kMainCt;
    V2_GLD16(W.w[0][0], voA, b, 0); V2_GLD16(W.w[0][1], voA, b, 1024);
  };
  auto load_x = [&](XSet& X, int cb, int q, int term) __attribute__((always_inline)) {
    const int voA = lane_now() << 4;
    const char* b = Wx + ((((int64_t)cb * NQ + q) * 2 + term) * 2) * kXCt;
    V2_GLD16(X.c[0], voA, b, 0); V2_GLD16(X.t[0], voA, b, 1024);
    V2_GLD16(X.c[1], voA, b, 2048); V2_GLD16(X.t[1], voA, b, 3072);
  };

  f32x4 acc[8][2];
#pragma unroll
  for (int g = 0; g < 8; ++g)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[g][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // LDS offsets of this lane's fragments inside a slab (16 rows further: + 2048, the swizzles repeat), recomputed per phase:
  //   main, tap t: row r = c16 + t, k chunk g4 at g4 ^ f1(r)
  //   cross, macro step q: K group g4 = tap 4 q + g4 -> row rx = c16 + 4 q + g4; chunk 4 at 4 ^ f2(rx); chunks 5, 6, 7: offset ^ 16, 32, 48
  auto main_off = [&](int l, int t) __attribute__((always_inline)) {
    const int r = (l & 15) + t;
    return r * V2_DROW + (((l >> 4) ^ (((r >> 1) & 3) << 1)) << 4);
  };
  auto cross_off = [&](int l, int q) __attribute__((always_inline)) {
    const int rx = (l & 15) + 4 * q + (l >> 4), px = rx >> 1;
    const int sx = (((px >> 1) & 1) << 2) | (((px >> 2) & 1) << 1) | (px & 1);
    return rx * V2_DROW + ((4 ^ sx) << 4);
  };

  // ---- a cross sub-phase: term 0 = w_lo6 x a_hi6 (chunks 4, 6), term 1 = w_hi6 x a_lo6 (chunks 5, 7); 16 scaled MFMAs
  auto x_phase = [&](const XSet& X, int cb, int q, int term) __attribute__((always_inline)) {
    const char* slab = smem + (cb & 1) * V2_DA_BYTES;
    const int ob = cross_off(lane_now(), q);
    const int oc = ob ^ (term << 4), ot = ob ^ (32 | (term << 4));
    v4i fc[NSLOT], ft[NSLOT];             // codes 0-15; {codes 16-23, scale dword (byte 0 hi, byte 1 lo), pad}: read NSLOT - 1 tiles ahead
    auto rd = [&](int g, int s) __attribute__((always_inline)) {
      fc[s] = *reinterpret_cast<const v4i*>(slab + oc + g * 2048);
      ft[s] = *reinterpret_cast<const v4i*>(slab + ot + g * 2048);
    };
#pragma unroll
    for (int g = 0; g < NSLOT - 1; ++g) rd(g, g);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int s = g % NSLOT;
      if (g + NSLOT - 1 < 8) rd(g + NSLOT - 1, (g + NSLOT - 1) % NSLOT);
      const v8i b = __builtin_shufflevector(fc[s], ft[s], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const v8i wa = __builtin_shufflevector(X.c[c], X.t[c], 0, 1, 2, 3, 4, 5, 6, 7);
        if (term == 0) acc[g][c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, b, acc[g][c], 2, 2, 0, X.t[c][2], 0, ft[s][2]);
        else           acc[g][c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, b, acc[g][c], 2, 2, 0, X.t[c][2], 1, ft[s][2]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // ---- a main phase: hi * hi of taps tap0 .. tap0 + NT - 1 (NT = 1, 2); 16 MFMAs per tap
  auto m_phase = [&](const WSet& W, int cb, int tap0, int nt) __attribute__((always_inline)) {
    const char* slab = smem + (cb & 1) * V2_DA_BYTES;
    int of[2];
    const int l = lane_now();
#pragma unroll
    for (int j = 0; j < 2; ++j) of[j] = j < nt ? main_off(l, tap0 + j) : 0;
    f16x8 fm[NSLOT][2];
    auto rd = [&](int g, int s) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (j < nt) fm[s][j] = *reinterpret_cast<const f16x8*>(slab + of[j] + g * 2048);
    };
#pragma unroll
    for (int g = 0; g < NSLOT - 1; ++g) rd(g, g);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int s = g % NSLOT;
      if (g + NSLOT - 1 < 8) rd(g + NSLOT - 1, (g + NSLOT - 1) % NSLOT);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (j < nt) {
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W.w[j][c], fm[s][j], acc[g][c], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  XSet XA, XB;
  WSet WA, WB;
  // prologue: slab cb_begin; XA, XB = the cross sets of (cb_begin, q 0); 7 taps: WA = taps 0, 1; 5 taps: WB = taps 2, 3
  for (int g = wave; g < NGRP; g += 4) dma_a(cb_begin, cb_begin & 1, g, lane_now());
  load_x(XA, cb_begin, 0, 0);
  load_x(XB, cb_begin, 0, 1);
  if (NTAPS == 7) load_w2(WA, cb_begin, 0);
  else load_w2(WB, cb_begin, 2);           // (5 and 9 taps issue WA = taps 0, 1 at the top of the channel block)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // VMEM issue order of one channel block (E = issued at the start of the phase) and the count each wait leaves in flight =
  // the operations issued behind the set it releases:
  //   7 taps  XA0: slab x5, WB(2,3) x4 | XB0: XA(q1) x4 | M01: XB(q1) x4 | M23: WA(4,5) x4 | XA1: WB(6) x2 | XB1: XA(q0') x4 | M45: XB(q0') x4 | M6: WA(0,1)' x4
  //   waits   XA0 8    XB0 13 (8) M01 13 (8) M23 8   XA1 8    XB1 6    M45 6 (2)    M6 8 (0)     (the last one stages no slab either)
  //   5 taps  XA0: WA(0,1) x4, slab x5 | XB0: XA(q1) x4 | M01: XB(q1) x4 | M23: WA(4) x2 | XA1: WB(2,3)' x4 | XB1: XA(q0') x4 | M4: XB(q0') x4
  //   waits   XA0 4    XB0 9 (4) M01 9 (4) M23 8    XA1 6    XB1 6 (2)    M4 8 (0)          (in brackets: the last channel block)
  // The last channel block issues nothing for a next one, so its final waits count fewer operations: it is a second, straight-line
  // copy of the body behind the loop (a run-time test inside one body splits it into blocks and costs the register allocation).
  auto body = [&](int cb, auto last_tag) __attribute__((always_inline)) {
    constexpr bool last = decltype(last_tag)::value;
    const int nx = cb + 1;
    if constexpr (NTAPS == 7) {
      V2_WAITX(8, XA);
      if constexpr (!last) dma_next(cb);
      load_w2(WB, cb, 2);
      __builtin_amdgcn_sched_barrier(0);
      x_phase(XA, cb, 0, 0);
      if constexpr (!last) V2_WAITX(13, XB); else V2_WAITX(8, XB);
      load_x(XA, cb, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      x_phase(XB, cb, 0, 1);
      if constexpr (!last) V2_WAITW2(13, WA); else V2_WAITW2(8, WA);
      load_x(XB, cb, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      m_phase(WA, cb, 0, 2);
      V2_WAITW2(8, WB);
      load_w2(WA, cb, 4);
      __builtin_amdgcn_sched_barrier(0);
      m_phase(WB, cb, 2, 2);
      V2_WAITX(8, XA);
      load_w1(WB, cb, 6);
      __builtin_amdgcn_sched_barrier(0);
      x_phase(XA, cb, 1, 0);
      V2_WAITX(6, XB);
      if constexpr (!last) load_x(XA, nx, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      x_phase(XB, cb, 1, 1);
      if constexpr (!last) { V2_WAITW2(6, WA); load_x(XB, nx, 0, 1); } else { V2_WAITW2(2, WA); }
      __builtin_amdgcn_sched_barrier(0);
      m_phase(WA, cb, 4, 2);
      if constexpr (!last) { V2_WAITW1(8, WB); load_w2(WA, nx, 0); } else { V2_WAITW1(0, WB); }
      __builtin_amdgcn_sched_barrier(0);
      m_phase(WB, cb, 6, 1);
    } else if constexpr (NTAPS == 9) {
      //   9 taps  XA0: WA(0,1) x4, slab x5 | XB0: XA(q1) x4 | M01: XB(q1) x4 | M23: WA(4,5) x4 | XA1: WB(6,7) x4 | XB1: XA(q2) x4 | M45: XB(q2) x4 |
      //           M67: WA(8) x2 | XA2: WB(2,3)' x4 | XB2: XA(q0') x4 | M8: XB(q0') x4
      //   waits   XA0 4   XB0 9 (4)   M01 9 (4)   M23 8   XA1 8   XB1 8   M45 8   M67 8   XA2 6   XB2 6 (2)   M8 8 (0)
      V2_WAITX(4, XA);
      load_w2(WA, cb, 0);
      if constexpr (!last) dma_next(cb);
      __builtin_amdgcn_sched_barrier(0);
      x_phase(XA, cb, 0, 0);
      if constexpr (!last) V2_WAITX(9, XB); else V2_WAITX(4, XB);
      load_x(XA, cb, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      x_phase(XB, cb, 0, 1);
      if constexpr (!last) V2_WAITW2(9, WA); else V2_WAITW2(4, WA);
      load_x(XB, cb, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      m_phase(WA, cb, 0, 2);
      V2_WAITW2(8, WB);
      load_w2(WA, cb, 4);
      __builtin_amdgcn_sched_barrier(0);
      m_phase(WB, cb, 2, 2);
      V2_WAITX(8, XA);
      load_w2(WB, cb, 6);
      __builtin_amdgcn_sched_barrier(0);
      x_phase(XA, cb, 1, 0);
      V2_WAITX(8, XB);
      load_x(XA, cb, 2, 0);
      __builtin_amdgcn_sched_barrier(0);
      x_phase(XB, cb, 1, 1);
      V2_WAITW2(8, WA);
      load_x(XB, cb, 2, 1);
      __builtin_amdgcn_sched_barrier(0);
      m_phase(WA, cb, 4, 2);
      V2_WAITW2(8, WB);
      load_w1(WA, cb, 8);
      __builtin_amdgcn_sched_barrier(0);
      m_phase(WB, cb, 6, 2);
      V2_WAITX(6, XA);
      if constexpr (!last) load_w2(WB, nx, 2);
      __builtin_amdgcn_sched_barrier(0);
      x_phase(XA, cb, 2, 0);
      if constexpr (!last) { V2_WAITX(6, XB); load_x(XA, nx, 0, 0); } else { V2_WAITX(2, XB); }
      __builtin_amdgcn_sched_barrier(0);
      x_phase(XB, cb, 2, 1);
      if constexpr (!last) { V2_WAITW1(8, WA); load_x(XB, nx, 0, 1); } else { V2_WAITW1(0, WA); }
      __builtin_amdgcn_sched_barrier(0);
      m_phase(WA, cb, 8, 1);
    } else {
      V2_WAITX(4, XA);
      load_w2(WA, cb, 0);                 // (the one set with a single macro step of lead: its registers serve tap 4 until the end of
                                          //  the channel block before; two cross sub-phases = 32 MFMAs of this wave, ~1.5 k cycles in wall time)
      if constexpr (!last) dma_next(cb);
      __builtin_amdgcn_sched_barrier(0);
      x_phase(XA, cb, 0, 0);
      if constexpr (!last) V2_WAITX(9, XB); else V2_WAITX(4, XB);
      load_x(XA, cb, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      x_phase(XB, cb, 0, 1);
      if constexpr (!last) V2_WAITW2(9, WA); else V2_WAITW2(4, WA);
      load_x(XB, cb, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      m_phase(WA, cb, 0, 2);
      V2_WAITW2(8, WB);
      load_w1(WA, cb, 4);
      __builtin_amdgcn_sched_barrier(0);
      m_phase(WB, cb, 2, 2);
      V2_WAITX(6, XA);
      if constexpr (!last) load_w2(WB, nx, 2);
      __builtin_amdgcn_sched_barrier(0);
      x_phase(XA, cb, 1, 0);
      if constexpr (!last) { V2_WAITX(6, XB); load_x(XA, nx, 0, 0); } else { V2_WAITX(2, XB); }
      __builtin_amdgcn_sched_barrier(0);
      x_phase(XB, cb, 1, 1);
      if constexpr (!last) { V2_WAITW1(8, WA); load_x(XB, nx, 0, 1); } else { V2_WAITW1(0, WA); }
      __builtin_amdgcn_sched_barrier(0);
      m_phase(WA, cb, 4, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                      // every wave is done reading slab cb; slab cb + 1 landed long ago (waits above)
  };
  for (int cb = cb_begin; cb + 1 < cb_end; ++cb) body(cb, std::false_type());
  // (hand-over to the straight-line copy: an empty asm over the accumulators ends their live ranges here, so that the allocator may
  // re-assign them for the copy instead of spilling one tile across the seam -- it did, in the 5-tap form)
#pragma unroll
  for (int g = 0; g < 8; ++g) asm volatile("" : "+v"(acc[g][0]), "+v"(acc[g][1]));
  body(cb_end - 1, std::true_type());
  // (nothing is in flight here: the last body waited for its last set with vmcnt(0) and its end-of-block barrier has been passed
  // by every wave, so the slab buffers may become the epilogue's scratch)
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  if (slice) {
    // K-split slice of a tail tile: the raw accumulators go to the slice's rows of GemmArgs::partial (p.Y / p.ldy were pointed there
    // by the kernel); 16-byte stores straight from the accumulator layout (lane & 15 -> frame, 4 * (lane >> 4) .. + 3 -> channels)
    const int c16e = lane_e & 15, g4e = lane_e >> 4;
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        *reinterpret_cast<f32x4*>(p.Y + (int64_t)(m0 + 16 * g + c16e) * p.ldy + n0 + wave * 32 + 16 * c + 4 * g4e) = acc[g][c];
    return;
  }
  // one epilogue per instantiation (both in one kernel cost 20 spilled accumulators at the loop exit)
  if constexpr (OUT_F6) store_wave_tile_n32_f6(p, acc, m0, n0 + wave * 32, lane_e, wave, smem);
  else store_wave_tile_n32<64, true>(p, acc, m0, n0 + wave * 32, lane_e, wave, smem);
}

// Workgroups [0, nMt * nNt) are whole tiles; with S > 0 the grid continues with the K-split slices of the tail tiles exactly as in
// gemm_bf16x3_w14p2_kernel (raw partial sums to GemmArgs::partial, finished by f6v2_tail_reduce_kernel).
template <int NTAPS, bool OUT_F6>
__global__ __launch_bounds__(256, 3) void gemm_f6v2_kernel(GemmArgs p, int nMt, int nNt, int S) {
  extern __shared__ __attribute__((aligned(16))) char smem_v2[];
  int mt, nt, cb_begin = 0, cb_end = p.cin >> 5;
  const bool slice = (int)blockIdx.x >= nMt * nNt;
  if (slice) {
    const int id = blockIdx.x - nMt * nNt;
    const int split = id % S, tile = id / S;
    mt = nMt + tile / nNt;
    nt = tile % nNt;
    const int per = cb_end / S;
    cb_begin = split * per;
    cb_end = cb_begin + per;
    p.Y = p.partial + ((int64_t)split * p.tail_mt - nMt) * (int64_t)V2_BM * p.Npad;   // row m of the tile -> slice row m - nMt * 128
    p.ldy = p.Npad;                       // (every row of a tail tile is stored; rows >= M are never read back)
  } else {
    const int tile = xcd_remap(blockIdx.x, nMt * nNt);
    mt = tile / nNt;
    nt = tile - mt * nNt;
  }
  f6v2_tile<NTAPS, OUT_F6>(p, mt * V2_BM, nt * V2_BN, smem_v2, cb_begin, cb_end, slice);
}

// One thread per (tail row, 32-channel block): ordered sum of the K slices, BN scale / shift + activation, then fp32 and / or the
// split-blocked row or the two-unit block (what the layer's whole tiles write through their epilogues).
template <int S>
__global__ __launch_bounds__(64) void f6v2_tail_reduce_kernel(GemmArgs p, int mt0) {
  const int nblk = p.Npad >> 5;
  const int64_t rows = (int64_t)p.tail_mt * V2_BM;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * nblk) return;
  const int64_t r = i / nblk;
  const int n0 = (int)(i - r * nblk) * 32;
  const int64_t m = (int64_t)mt0 * V2_BM + r;
  if (m >= p.M) return;
  bool zero;
  const int orow = out_row(p, (int)m, zero);
  if (orow < 0) return;
  float v[32];
#pragma unroll
  for (int e = 0; e < 32; ++e) v[e] = 0.f;
  // S is a template parameter: the S x 8 loads of a thread are issued together (as a run-time loop they were S dependent
  // round trips: 14-16 us per layer for 1.4 % of its work); the slices are still added in ascending order
  f32x4 t[S][8];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const float* src = p.partial + ((int64_t)s * rows + r) * p.Npad + n0;
#pragma unroll
    for (int q = 0; q < 8; ++q) t[s][q] = *reinterpret_cast<const f32x4*>(src + 4 * q);
  }
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int q = 0; q < 8; ++q) { v[4 * q] += t[s][q][0]; v[4 * q + 1] += t[s][q][1]; v[4 * q + 2] += t[s][q][2]; v[4 * q + 3] += t[s][q][3]; }
#pragma unroll
  for (int e = 0; e < 32; ++e) {
    const int n = n0 + e;
    v[e] = (n < p.N && !zero) ? apply_act(fmaf(v[e], p.scale[n], p.shift[n]), p.act, p.alpha ? p.alpha[n] : 0.f) : 0.f;
  }
  if (p.Y && n0 < p.N) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (n0 + 4 * q < p.N) *reinterpret_cast<f32x4*>(p.Y + (int64_t)orow * p.ldy + n0 + 4 * q) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
  }
  if (!p.Ysb || n0 >= p.ldsb) return;
  uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<char*>(p.Ysb) + (int64_t)orow * p.ldsb * 4 + (n0 >> 5) * 128);
  uint32_t hw[16], lw[16];
  float hf[32], lf[32], mh = 0.f, ml = 0.f, mx = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float a = v[2 * e] * p.sb_mul, b = v[2 * e + 1] * p.sb_mul;
    split2t<true>(a, b, hw[e], lw[e]);
    const f16x2_t hh = __builtin_bit_cast(f16x2_t, hw[e]);
    hf[2 * e] = (float)hh[0]; hf[2 * e + 1] = (float)hh[1];
    lf[2 * e] = a - hf[2 * e]; lf[2 * e + 1] = b - hf[2 * e + 1];
    mh = fmaxf(mh, fmaxf(fabsf(hf[2 * e]), fabsf(hf[2 * e + 1])));
    ml = fmaxf(ml, fmaxf(fabsf(lf[2 * e]), fabsf(lf[2 * e + 1])));
    mx = fmaxf(mx, fmaxf(fabsf(a), fabsf(b)));
  }
  ovf_report(p.ovf, mx);
#pragma unroll
  for (int c = 0; c < 4; ++c) dst[c] = uint4{hw[4 * c], hw[4 * c + 1], hw[4 * c + 2], hw[4 * c + 3]};
  if (!p.ysb_f6) {
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[4 + c] = uint4{lw[4 * c], lw[4 * c + 1], lw[4 * c + 2], lw[4 * c + 3]};
    return;
  }
  uint32_t bh, bl;
  const float ih = e8m0_of(mh, bh), il = e8m0_of(ml, bl);
  uint32_t ch[6] = {0, 0, 0, 0, 0, 0}, cl[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const uint32_t a = e2m3_code(hf[k], ih), b = e2m3_code(lf[k], il);
    const int bit = 6 * k, w = bit >> 5, sh = bit & 31;
    ch[w] |= a << sh; cl[w] |= b << sh;
    if (sh > 26) { ch[w + 1] |= a >> (32 - sh); cl[w + 1] |= b >> (32 - sh); }
  }
  const uint32_t sc2 = bh | (bl << 8);
  dst[4] = uint4{ch[0], ch[1], ch[2], ch[3]};
  dst[5] = uint4{cl[0], cl[1], cl[2], cl[3]};
  dst[6] = uint4{ch[4], ch[5], sc2, 0u};
  dst[7] = uint4{cl[4], cl[5], sc2, 0u};
  if (m == p.M - 1) {                      // eight zero rows behind the value's last row (store_wave_tile_n32_f6)
    for (int k = 1; k <= 8; ++k)
#pragma unroll
      for (int c = 0; c < 8; ++c) dst[(int64_t)k * (p.ldsb >> 2) + c] = uint4{0u, 0u, 0u, 0u};
  }
}

#undef V2_GLD16
#undef V2_WAITX
#undef V2_WAITW2
#undef V2_WAITW1

// a.Xsb = activations in the two-unit block format (row stride a.ldsbx channels), a.Wfr / a.Wx6 = main / cross weights
// (xvec_api.hip, upload_layer), a.K = taps * a.cin, taps 5, 7 or 9, a.cin % 32 == 0.  a.tail_mt / a.ksplit / a.partial: the K-split tail
// (gemm_bf16x3_tail_plan), decided at plan time.
hipError_t launch_gemm_f16f6(const GemmArgs& a, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  const int taps = a.cin > 0 ? a.K / a.cin : 0;
  // Only the widths the reference's graphs contain are instantiated (tdnn: 5, 5, 7; extended tdnn: 5, 5, 7, 9).
  if ((taps != 5 && taps != 7 && taps != 9) || (a.cin & 31) || a.ldsbx != a.cin || !a.Wx6 || !a.Wfr || a.a_pitch || a.pool_part || a.R || (a.N & 3))
    return hipErrorInvalidValue;
  static std::mutex mu;
  static bool attr_set[64] = {};
  const size_t smem = (size_t)2 * v2_slab_bytes(taps);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (!attr_set[dev & 63]) {
      const void* ks[] = {reinterpret_cast<const void*>(gemm_f6v2_kernel<5, false>), reinterpret_cast<const void*>(gemm_f6v2_kernel<7, false>),
                          reinterpret_cast<const void*>(gemm_f6v2_kernel<5, true>), reinterpret_cast<const void*>(gemm_f6v2_kernel<7, true>),
                          reinterpret_cast<const void*>(gemm_f6v2_kernel<9, false>), reinterpret_cast<const void*>(gemm_f6v2_kernel<9, true>)};
      for (const void* k : ks) {
        const hipError_t r = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * v2_slab_bytes(9));
        if (r != hipSuccess) return r;
      }
      attr_set[dev & 63] = true;
    }
  }
  const int nMt = (a.M + V2_BM - 1) / V2_BM, nNt = a.Npad / V2_BN, ncb = a.cin >> 5;
  const bool tail = a.tail_mt > 0 && (a.ksplit == 2 || a.ksplit == 4 || a.ksplit == 8) && a.partial && a.tail_mt < nMt && ncb % a.ksplit == 0;
  const int nMain = tail ? nMt - a.tail_mt : nMt;
  const int S = tail ? a.ksplit : 0;
  const dim3 grid(nMain * nNt + (tail ? a.tail_mt * nNt * a.ksplit : 0)), block(256);
  const bool out_f6 = a.ysb_f6 && a.Ysb && !a.Y;
  if (a.ysb_f6 && !out_f6) return hipErrorInvalidValue;        // the block format is written only when it is the layer's one output
  if (taps == 5) {
    if (out_f6) hipLaunchKernelGGL((gemm_f6v2_kernel<5, true>), grid, block, smem, s, a, nMain, nNt, S);
    else        hipLaunchKernelGGL((gemm_f6v2_kernel<5, false>), grid, block, smem, s, a, nMain, nNt, S);
  } else if (taps == 7) {
    if (out_f6) hipLaunchKernelGGL((gemm_f6v2_kernel<7, true>), grid, block, smem, s, a, nMain, nNt, S);
    else        hipLaunchKernelGGL((gemm_f6v2_kernel<7, false>), grid, block, smem, s, a, nMain, nNt, S);
  } else {
    if (out_f6) hipLaunchKernelGGL((gemm_f6v2_kernel<9, true>), grid, block, smem, s, a, nMain, nNt, S);
    else        hipLaunchKernelGGL((gemm_f6v2_kernel<9, false>), grid, block, smem, s, a, nMain, nNt, S);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || !tail) return e;
  return launch_f6v2_tail_reduce(a, nMain, s);
}

// The reduce of a K-split tail (rows from M tile nMain on; a.partial, a.tail_mt, a.ksplit as planned) with the split-blocked or the
// two-unit block output; also the tail of a one-tap or f16 multi-tap layer that writes the block format (gemm_bf16x3.hip).
hipError_t launch_f6v2_tail_reduce(const GemmArgs& a, int nMain, hipStream_t s) {
  const int64_t total = (int64_t)a.tail_mt * V2_BM * (a.Npad >> 5);
  const dim3 rgrid((unsigned)((total + 63) / 64)), rblock(64);
  switch (a.ksplit) {
    case 2: hipLaunchKernelGGL(f6v2_tail_reduce_kernel<2>, rgrid, rblock, 0, s, a, nMain); break;
    case 4: hipLaunchKernelGGL(f6v2_tail_reduce_kernel<4>, rgrid, rblock, 0, s, a, nMain); break;
    case 8: hipLaunchKernelGGL(f6v2_tail_reduce_kernel<8>, rgrid, rblock, 0, s, a, nMain); break;
    default: return hipErrorInvalidValue;             // gemm_bf16x3_tail_plan picks 2, 4 or 8 slices
  }
  return hipGetLastError();
}

}  // namespace xv
