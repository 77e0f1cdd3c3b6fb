// Two-unit split-precision GEMM (XV_PREC_F16F6) for the 5-, 7- and 9-tap temporal convolutions of the (extended) TDNN and, as a
// three-tap form, the stride-1 3 x 3 convolutions of the ResNet stages.  Three workgroups per CU, every load two or three phases
// ahead of its use, counted waits only.  The arithmetic:
//
//   a * w ~ f16(a) * f16(w)                                                   v_mfma_f32_16x16x32_f16
//         + q6(f16(a)) * q6(w - f16(w)) + q6(a - f16(a)) * q6(f16(w))          2 x v_mfma_scale_f32_16x16x128_f8f6f4 (e2m3, E8M0 block scale)
//
// Activations in the block format of gemm_f16f6.hip (128 B per (row, 32 channels): f16 hi | fp6 codes of hi and lo | chunks 6 / 7 =
// [8 B code tail | scale dword | pad] of q6(hi) / q6(lo)), staged slab by slab (128 frames + halo x 32 channels) by LDS-DMA; 128 x 128
// workgroup tile of four waves, each wave one 32-channel block x 128 frames (64 accumulator registers); weights from L2 straight
// into registers in MFMA-fragment order.
//
// The K = 128 of a scaled MFMA is four "K groups" of 32 channels; a K group is one (channel block, tap) pair -- lane group l >> 4 reads
// slab row frame + tap of its pair's block.  Pairs are enumerated block-major across a QUAD of channel blocks: 4 T pairs = T macro
// steps exactly, whatever T (one block at a time would need ceil(T / 4) steps per block with zero-weight groups in the last: 25 %
// more cross MFMAs for T = 3 and 9, 60 % for T = 5, 14 % for T = 7).  A macro step then spans at most two neighbouring blocks; it
// runs during the later one, with both slabs resident: three slab buffers in a ring (52 KB, still three workgroups per CU), slab
// c + 1 issued at the top of block c into the buffer slab c - 2 left at the barrier before.
//
// A quad is a list of PHASES (V2Q<T>::ph below): a cross phase = one term of a macro step (16 scaled MFMAs, one 16-register operand
// set), a main phase = hi * hi of one or two taps (16 / 32 MFMAs, 8 / 16 registers).  Sets rotate: the set a phase has consumed is
// reloaded at the start of the next phase for the phase NS - 1 further on, so every load has NS - 1 phases (32 - 96 MFMAs of this
// wave, three times that in wall time) to land, and every s_waitcnt vmcnt(N) is computed from the table at compile time (v2q_wait:
// the loads and LDS-DMA pieces issued behind the awaited set) -- no hand-counted tables, no drain.  NS = 3 sets (48 registers) for 5,
// 7, 9 taps, 4 for the short three-tap body; with the accumulators and two fragment slots 164 - 168 VGPRs.  The last quad of a tile
// is a second, straight-line copy that issues nothing for a next one (a run-time test inside the body would split it into blocks
// and cost the register allocation); per-lane LDS / DMA offsets are recomputed from the lane id per phase (selects, never
// branches) instead of living in loop-invariant registers.  History: round 2 ran two macro steps of main weights + cross weights
// at 250 VGPRs = two workgroups per CU with a vmcnt(0) drain per macro step (54 % matrix-pipe busy); round 3 first hand-scheduled
// eight phases per channel block on four named sets (63 %), then this generated form, which removed the zero K groups.
#include <mutex>
#include <type_traits>
#include <utility>

#include "xv_f6.h"

namespace xv {

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr2_t;

constexpr int V2_BM = 128, V2_BN = 128, V2_DROW = 128;
// slab rows: 128 frames + the halo of the last tap (a K group of the cross terms reads row frame + its tap, like the main pass), in
// whole 8-row DMA pieces: 136 rows for up to nine taps
constexpr int kV2Groups = 17, kV2SlabBytes = kV2Groups * 8 * V2_DROW, kV2Ring = 3;      // 17 408 B per slab, 52 224 B of LDS
constexpr int64_t kMainCt = 64 * 16;               // bytes per (tap, 16-channel tile) of the main weights: 64 lanes x 16 B
constexpr int64_t kXCt = 2 * 64 * 16;              // per (macro step, term, 16-channel tile) of the cross weights: 64 x 16 B codes 0-15 |
                                                   // 64 x 16 B {codes 16-23, scale dword, pad}: one lane offset serves both planes

// Generated schedules.  The 3-tap (ResNet) and the 7-tap form run a QUAD of channel blocks per loop body: 4 T (block, tap) pairs
// fill T macro steps of the cross terms exactly (one block at a time needs ceil(T / 4) steps of four K groups per block: a zero K
// group in every block for T = 3 -- 80 MFMAs per block instead of 72 --, one in every second step for T = 7: 176 instead of 168).
// A body is a list of PHASES, each with its own register set, in a fixed rotation of four sets: the set a phase has just
// consumed is reloaded, at the start of the next phase, for the phase three further on -- every load has three phases (48 - 96
// MFMAs of this wave) to land -- and every wait count follows from the table (v2q_wait).  kind 0 = hi * hi of `b` taps from tap `a`
// of block `blk`; kind 1 = cross term `b` of macro step `a`, run in the block whose slab completes its K groups (step q covers pairs
// 4 q .. 4 q + 3 = blocks (4 q) / T .. (4 q + 3) / T, at most two; both slabs are resident: ring of three buffers).  Some two-tap
// phases are split into single taps so that the phase count is a multiple of the four sets.
// A phase runs during block `blk` of the quad (the slabs of blocks blk - 1 and blk are resident then).  kind 1: a = macro step,
// b = term.  kind 0: tap b of block a and, unless c < 0, tap d of block c (a, c in {blk - 1, blk}).
struct V2Phase { int kind, blk, a, b, c, d; };
template <int T> struct V2Q;
template <> struct V2Q<3> {
  static constexpr int NPH = 16, NX = 3, NS = 4;
  static constexpr V2Phase ph[NPH] = {{0, 0, 0, 0, 0, 1}, {0, 0, 0, 2, -1, 0},
                                      {1, 1, 0, 0, 0, 0}, {1, 1, 0, 1, 0, 0}, {0, 1, 1, 0, 1, 1}, {0, 1, 1, 2, -1, 0},
                                      {1, 2, 1, 0, 0, 0}, {1, 2, 1, 1, 0, 0}, {0, 2, 2, 0, -1, 0}, {0, 2, 2, 1, -1, 0}, {0, 2, 2, 2, -1, 0},
                                      {1, 3, 2, 0, 0, 0}, {1, 3, 2, 1, 0, 0}, {0, 3, 3, 0, -1, 0}, {0, 3, 3, 1, -1, 0}, {0, 3, 3, 2, -1, 0}};
};
// 7 taps: thirty phases on a rotation of THREE sets (a load has two phases = 32 - 64 MFMAs to land, the shortest lead of the
// hand-written bodies; a fourth set does not fit beside the 64 accumulators and two fragment slots of this body's phases).
//   steps: q0 = block 0 taps 0-3 | q1 = b0 t4-6, b1 t0 | q2 = b1 t1-4 | q3 = b1 t5-6, b2 t0-1 | q4 = b2 t2-5 | q5 = b2 t6, b3 t0-2 | q6 = b3 t3-6
template <> struct V2Q<7> {
  static constexpr int NPH = 30, NX = 7, NS = 3;
  static constexpr V2Phase ph[NPH] = {
      {1, 0, 0, 0, 0, 0}, {1, 0, 0, 1, 0, 0}, {0, 0, 0, 0, 0, 1}, {0, 0, 0, 2, 0, 3}, {0, 0, 0, 4, 0, 5}, {0, 0, 0, 6, -1, 0},
      {1, 1, 1, 0, 0, 0}, {1, 1, 1, 1, 0, 0}, {0, 1, 1, 0, 1, 1}, {0, 1, 1, 2, 1, 3}, {1, 1, 2, 0, 0, 0}, {1, 1, 2, 1, 0, 0}, {0, 1, 1, 4, 1, 5}, {0, 1, 1, 6, -1, 0},
      {1, 2, 3, 0, 0, 0}, {1, 2, 3, 1, 0, 0}, {0, 2, 2, 0, 2, 1}, {0, 2, 2, 2, 2, 3}, {1, 2, 4, 0, 0, 0}, {1, 2, 4, 1, 0, 0}, {0, 2, 2, 4, 2, 5}, {0, 2, 2, 6, -1, 0},
      {1, 3, 5, 0, 0, 0}, {1, 3, 5, 1, 0, 0}, {0, 3, 3, 0, 3, 1}, {0, 3, 3, 2, 3, 3}, {1, 3, 6, 0, 0, 0}, {1, 3, 6, 1, 0, 0}, {0, 3, 3, 4, 3, 5}, {0, 3, 3, 6, -1, 0}};
};
// 5 taps: 20 pairs = five macro steps (one block at a time: eight slots per block, three of them zero: 144 MFMAs per block instead
// of 120).  q0 = b0 t0-3 | q1 = b0 t4, b1 t0-2 | q2 = b1 t3-4, b2 t0-1 | q3 = b2 t2-4, b3 t0 | q4 = b3 t1-4.  24 phases on three sets.
template <> struct V2Q<5> {
  static constexpr int NPH = 24, NX = 5, NS = 3;
  static constexpr V2Phase ph[NPH] = {
      {1, 0, 0, 0, 0, 0}, {1, 0, 0, 1, 0, 0}, {0, 0, 0, 0, -1, 0}, {0, 0, 0, 1, -1, 0}, {0, 0, 0, 2, 0, 3}, {0, 0, 0, 4, -1, 0},
      {1, 1, 1, 0, 0, 0}, {1, 1, 1, 1, 0, 0}, {0, 1, 1, 0, -1, 0}, {0, 1, 1, 1, -1, 0}, {0, 1, 1, 2, 1, 3}, {0, 1, 1, 4, -1, 0},
      {1, 2, 2, 0, 0, 0}, {1, 2, 2, 1, 0, 0}, {0, 2, 2, 0, 2, 1}, {0, 2, 2, 2, 2, 3}, {0, 2, 2, 4, -1, 0},
      {1, 3, 3, 0, 0, 0}, {1, 3, 3, 1, 0, 0}, {0, 3, 3, 0, 3, 1}, {0, 3, 3, 2, 3, 3}, {1, 3, 4, 0, 0, 0}, {1, 3, 4, 1, 0, 0}, {0, 3, 3, 4, -1, 0}};
};
// 9 taps: 36 pairs = nine macro steps (one block at a time: twelve slots per block, three of them zero: 240 MFMAs per block
// instead of 216).  q0 = b0 t0-3 | q1 = b0 t4-7 | q2 = b0 t8, b1 t0-2 | q3 = b1 t3-6 | q4 = b1 t7-8, b2 t0-1 | q5 = b2 t2-5 |
// q6 = b2 t6-8, b3 t0 | q7 = b3 t1-4 | q8 = b3 t5-8.  39 phases on three sets.
template <> struct V2Q<9> {
  static constexpr int NPH = 39, NX = 9, NS = 3;
  static constexpr V2Phase ph[NPH] = {
      {1, 0, 0, 0, 0, 0}, {1, 0, 0, 1, 0, 0}, {0, 0, 0, 0, 0, 1}, {0, 0, 0, 2, 0, 3}, {1, 0, 1, 0, 0, 0}, {1, 0, 1, 1, 0, 0}, {0, 0, 0, 4, 0, 5}, {0, 0, 0, 6, 0, 7}, {0, 0, 0, 8, -1, 0},
      {1, 1, 2, 0, 0, 0}, {1, 1, 2, 1, 0, 0}, {0, 1, 1, 0, 1, 1}, {0, 1, 1, 2, 1, 3}, {1, 1, 3, 0, 0, 0}, {1, 1, 3, 1, 0, 0}, {0, 1, 1, 4, 1, 5}, {0, 1, 1, 6, 1, 7}, {0, 1, 1, 8, -1, 0},
      {1, 2, 4, 0, 0, 0}, {1, 2, 4, 1, 0, 0}, {0, 2, 2, 0, 2, 1}, {0, 2, 2, 2, 2, 3}, {1, 2, 5, 0, 0, 0}, {1, 2, 5, 1, 0, 0}, {0, 2, 2, 4, 2, 5}, {0, 2, 2, 6, 2, 7}, {0, 2, 2, 8, -1, 0},
      {1, 3, 6, 0, 0, 0}, {1, 3, 6, 1, 0, 0}, {0, 3, 3, 0, -1, 0}, {0, 3, 3, 1, -1, 0}, {0, 3, 3, 2, 3, 3}, {1, 3, 7, 0, 0, 0}, {1, 3, 7, 1, 0, 0}, {0, 3, 3, 4, 3, 5},
      {1, 3, 8, 0, 0, 0}, {1, 3, 8, 1, 0, 0}, {0, 3, 3, 6, 3, 7}, {0, 3, 3, 8, -1, 0}};
};
template <class Q> constexpr int v2q_mod(int i) { return ((i % Q::NPH) + Q::NPH) % Q::NPH; }
template <class Q> constexpr int v2q_loads(int i) { return Q::ph[v2q_mod<Q>(i)].kind == 1 || Q::ph[v2q_mod<Q>(i)].c >= 0 ? 4 : 2; }
template <class Q> constexpr bool v2q_first(int i) { return v2q_mod<Q>(i) == 0 || Q::ph[v2q_mod<Q>(i)].blk != Q::ph[v2q_mod<Q>(i) - 1].blk; }
template <class Q> constexpr bool v2q_end(int i) { return v2q_mod<Q>(i) == Q::NPH - 1 || Q::ph[v2q_mod<Q>(i)].blk != Q::ph[v2q_mod<Q>(i) + 1].blk; }
// does phase j (relative to this quad; negative = the quad before, which is never the last) issue a set / a slab?
template <class Q> constexpr bool v2q_has_set(int j, bool last) { return j + Q::NS - 1 < Q::NPH || !last; }
template <class Q> constexpr bool v2q_has_slab(int j, bool last) { return v2q_first<Q>(j) && (j < 0 || !(last && Q::ph[j].blk == 3)); }
// operations in flight behind the set of phase i when it is awaited: the issues of phases i - (NS - 1) (its slab only) .. i - 1,
// each "set for the phase NS - 1 on, then the slab of the next block if the phase opens a block"
template <class Q> constexpr int v2q_wait(int i, bool last) {
  int n = 0;
  for (int j = i - (Q::NS - 1); j <= i - 1; ++j) {
    if (j != i - (Q::NS - 1) && v2q_has_set<Q>(j, last)) n += v2q_loads<Q>(j + Q::NS - 1);
    if (v2q_has_slab<Q>(j, last)) n += 5;
  }
  return n;
}
// before the barrier that ends a block: everything issued behind its slab = the sets issued by its later phases
template <class Q> constexpr int v2q_wait_slab(int iend, bool last) {
  int n = 0;
  for (int j = iend; j >= 0 && !v2q_first<Q>(j); --j)
    if (v2q_has_set<Q>(j, last)) n += v2q_loads<Q>(j + Q::NS - 1);
  return n;
}
// A table is complete and consistent: every (block, tap) of the quad in exactly one main phase, every (macro step, term) in exactly
// one cross phase -- in the block that holds the last of its four pairs --, a phase only touches the block it runs in or the one
// before (the two resident slabs), blocks in order, and the phase count a multiple of the set count (the rotation closes).
template <class Q> constexpr bool v2q_valid(int T) {
  if (Q::NPH % Q::NS != 0 || Q::NX != T) return false;
  for (int b = 0; b < 4; ++b)
    for (int t = 0; t < T; ++t) {
      int n = 0;
      for (int i = 0; i < Q::NPH; ++i) {
        const V2Phase& p = Q::ph[i];
        if (p.kind != 0) continue;
        if (p.a == b && p.b == t) ++n;
        if (p.c == b && p.d == t) ++n;
      }
      if (n != 1) return false;
    }
  for (int q = 0; q < T; ++q)
    for (int term = 0; term < 2; ++term) {
      int n = 0;
      for (int i = 0; i < Q::NPH; ++i) {
        const V2Phase& p = Q::ph[i];
        if (p.kind == 1 && p.a == q && p.b == term) { ++n; if (p.blk != (4 * q + 3) / T) return false; }
      }
      if (n != 1) return false;
    }
  for (int i = 0; i < Q::NPH; ++i) {
    const V2Phase& p = Q::ph[i];
    if (p.blk < 0 || p.blk > 3 || (i > 0 && p.blk < Q::ph[i - 1].blk)) return false;
    if (p.kind == 0) {
      if (p.a != p.blk && p.a != p.blk - 1) return false;
      if (p.c >= 0 && p.c != p.blk && p.c != p.blk - 1) return false;
      if (p.a < 0 || p.b < 0 || p.b >= T || (p.c >= 0 && (p.d < 0 || p.d >= T))) return false;
    }
  }
  return true;
}
static_assert(v2q_valid<V2Q<3>>(3) && v2q_valid<V2Q<5>>(5) && v2q_valid<V2Q<7>>(7) && v2q_valid<V2Q<9>>(9), "phase table");

template <class F, int... I>
__device__ __forceinline__ void v2q_for(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>()), ...); }

}  // namespace

#define V2_GLD16(dst, voff, sbase, OFF) asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #OFF : "=v"(dst) : "v"(voff), "s"(sbase))
template <int NTAPS, bool OUT_F6, bool RES>
__device__ __forceinline__ void f6v2_tile(const GemmArgs& p, int m0, int n0, char* smem, int cb_begin, int cb_end, bool slice, int bin = 0) {
  static_assert(NTAPS == 3 || NTAPS == 5 || NTAPS == 7 || NTAPS == 9, "taps");
  constexpr int NQ = (NTAPS + 3) / 4;    // (the main weights are stored in 4 NQ tap slots per channel block)
  constexpr int NGRP = kV2Groups, V2_DA_BYTES = kV2SlabBytes;
  constexpr int NSLOT = 2;               // fragment slots per phase: tile g + 1 is read while tile g multiplies
  constexpr int RING = kV2Ring;          // slab buffers: a macro step spans two channel blocks, both stay resident
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
  const char* Abase = reinterpret_cast<const char*>(p.Xsb) + (int64_t)m0 * a_row_bytes + (NTAPS == 3 ? (int64_t)bin * p.bin_x_bytes : 0);
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
  auto buf_of = [&](int cb) __attribute__((always_inline)) { return cb % RING; };
  auto dma_next = [&](int cb) __attribute__((always_inline)) {  // slab cb + 1: five pieces per wave (17 groups, the last ones duplicates)
    const int nx = cb + 1;
    const int l = lane_now();
    const int nb_ = buf_of(nx);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int g = wave + 4 * i;
      dma_a(nx, nb_, g < NGRP ? g : NGRP - 1, l);
    }
  };
  // weights of this wave's 32-channel block: uniform bases (scalar registers) + one lane offset each
  const int nb = (n0 >> 5) + wave;
  const char* Wm = reinterpret_cast<const char*>(p.Wfr) + (int64_t)nb * ncb * (8 * NQ) * kMainCt;     // [cb][4 NQ taps][2 tiles][1 KB]
  // cross weights: [cb][q][term][2 tiles][2 KB]; 5 taps: [pair of channel blocks][3 macro steps][term][2 tiles][2 KB]
  //                3 taps: [quad of channel blocks][3 macro steps][term][2 tiles][2 KB]
  //                7 taps: [quad][7 macro steps][term][2 tiles][2 KB]
  const char* Wx = reinterpret_cast<const char*>(p.Wx6) + (int64_t)nb * (ncb >> 2) * (4 * NTAPS) * kXCt;

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

  // (row c16 + tap of slab buffer `b`, as an offset from the start of the LDS: the buffer bases have bits 4, 5 clear)
  auto cross_at = [&](int l, int base, int tap) __attribute__((always_inline)) {      // base = byte offset of the slab buffer
    const int rx = (l & 15) + tap, px = rx >> 1;
    const int sx = (((px >> 1) & 1) << 2) | (((px >> 2) & 1) << 1) | (px & 1);
    return base + rx * V2_DROW + ((4 ^ sx) << 4);
  };
  {
    // ---- quads of channel blocks on the generated schedule V2Q<NTAPS> (3 taps = the ResNet form: 3 x 3 on the zero-bordered grid
    //      as three taps along time over the 3 C channels of a kernel row)
    using Q = V2Q<NTAPS>;
    struct RSet { v4i r[4]; };             // a cross set {codes 0-15, tail | scale} x 2 channel tiles, or main weights [tap][tile]
    RSet S[Q::NS];
    auto load_set = [&](RSet& R, int cbq, auto idx) __attribute__((always_inline)) {   // the set of phase idx of the quad at cbq
      constexpr V2Phase ph = Q::ph[v2q_mod<Q>(decltype(idx)::value)];
      const int voA = lane_now() << 4;
      if constexpr (ph.kind == 1) {
        const char* b = Wx + ((((int64_t)(cbq >> 2) * Q::NX + ph.a) * 2 + ph.b) * 2) * kXCt;
        V2_GLD16(R.r[0], voA, b, 0); V2_GLD16(R.r[1], voA, b, 1024);
        V2_GLD16(R.r[2], voA, b, 2048); V2_GLD16(R.r[3], voA, b, 3072);
      } else {
        const char* b = Wm + ((int64_t)(cbq + ph.a) * (8 * NQ) + ph.b * 2) * kMainCt;
        V2_GLD16(R.r[0], voA, b, 0); V2_GLD16(R.r[1], voA, b, 1024);
        if constexpr (ph.c >= 0) {
          const char* b2 = Wm + ((int64_t)(cbq + ph.c) * (8 * NQ) + ph.d * 2) * kMainCt;
          V2_GLD16(R.r[2], voA, b2, 0); V2_GLD16(R.r[3], voA, b2, 1024);
        }
      }
    };
    // ---- a cross phase: term 0 = w_lo6 x a_hi6 (chunks 4, 6), term 1 = w_hi6 x a_lo6 (chunks 5, 7); 16 scaled MFMAs;
    //      ob = this lane's K group: LDS offset of chunk 4 of its row in tile 0; fragments read NSLOT - 1 tiles ahead
    auto x_phase_s = [&](const RSet& X, int ob, int term) __attribute__((always_inline)) {
      const int oc = ob ^ (term << 4), ot = ob ^ (32 | (term << 4));
      v4i fc[NSLOT], ft[NSLOT];
      auto rd = [&](int g, int s) __attribute__((always_inline)) {
        fc[s] = *reinterpret_cast<const v4i*>(smem + oc + g * 2048);
        ft[s] = *reinterpret_cast<const v4i*>(smem + ot + g * 2048);
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
          const v8i wa = __builtin_shufflevector(X.r[2 * c], X.r[2 * c + 1], 0, 1, 2, 3, 4, 5, 6, 7);
          if (term == 0) acc[g][c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, b, acc[g][c], 2, 2, 0, X.r[2 * c + 1][2], 0, ft[s][2]);
          else           acc[g][c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, b, acc[g][c], 2, 2, 0, X.r[2 * c + 1][2], 1, ft[s][2]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // ---- a main phase: hi * hi of one or two taps (possibly of neighbouring blocks); 16 MFMAs per tap
    auto m_phase_s = [&](const RSet& W, int cbA, int tapA, int cbB, int tapB, int nt) __attribute__((always_inline)) {
      const char* slab = smem;
      int of[2];
      const int l = lane_now();
      of[0] = buf_of(cbA) * V2_DA_BYTES + main_off(l, tapA);
      of[1] = nt > 1 ? buf_of(cbB) * V2_DA_BYTES + main_off(l, tapB) : 0;
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
            for (int c = 0; c < 2; ++c)
              acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W.r[2 * j + c]), fm[s][j], acc[g][c], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // K groups of macro step q of the quad at cbq: pair 4 q + g = block (4 q + g) / T, tap (4 q + g) % T -- at most two neighbouring
    // blocks, chosen by a select
    auto ob_of = [&](int cbq, int q) __attribute__((always_inline)) {
      const int l = lane_now(), pr = 4 * q + (l >> 4);
      const int lo = (4 * q) / NTAPS;
      const int hi_sel = pr >= NTAPS * (lo + 1);
      const int tap = pr - NTAPS * (lo + hi_sel);
      const int b_lo = buf_of(cbq + lo) * V2_DA_BYTES, b_hi = buf_of(cbq + lo + 1) * V2_DA_BYTES;
      return cross_at(l, hi_sel ? b_hi : b_lo, tap);
    };
    // prologue: slab cb_begin, the sets of phases 0, 1, 2
    for (int g = wave; g < NGRP; g += 4) dma_a(cb_begin, buf_of(cb_begin), g, lane_now());
    load_set(S[0], cb_begin, std::integral_constant<int, 0>());
    load_set(S[1], cb_begin, std::integral_constant<int, 1>());
    if constexpr (Q::NS == 4) load_set(S[2], cb_begin, std::integral_constant<int, 2>());
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto quad = [&](int cbq, auto last_tag) __attribute__((always_inline)) {
      constexpr bool last = decltype(last_tag)::value;
      auto phase = [&](auto idx) __attribute__((always_inline)) {
        constexpr int i = decltype(idx)::value;
        constexpr V2Phase ph = Q::ph[i];
        RSet& R = S[i % Q::NS];
        if constexpr (ph.kind == 0 && ph.c < 0) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(R.r[0]), "+v"(R.r[1]) : "n"(v2q_wait<Q>(i, last)));
        else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(R.r[0]), "+v"(R.r[1]), "+v"(R.r[2]), "+v"(R.r[3]) : "n"(v2q_wait<Q>(i, last)));
        if constexpr (v2q_has_set<Q>(i, last))
          load_set(S[(i + Q::NS - 1) % Q::NS], i + Q::NS - 1 < Q::NPH ? cbq : cbq + 4, std::integral_constant<int, (i + Q::NS - 1) % Q::NPH>());
        if constexpr (v2q_has_slab<Q>(i, last)) dma_next(cbq + ph.blk);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ph.kind == 1) x_phase_s(R, ob_of(cbq, ph.a), ph.b);
        else m_phase_s(R, cbq + ph.a, ph.b, cbq + (ph.c < 0 ? ph.a : ph.c), ph.d, ph.c < 0 ? 1 : 2);
        if constexpr (v2q_end<Q>(i)) {
          // the block's slab pieces (issued by its first phase) have landed before the barrier hands the buffer over
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(v2q_wait_slab<Q>(i, last)) : "memory");
          __builtin_amdgcn_sched_barrier(0);
          __syncthreads();                // every wave is done with the slab two blocks back; the next one is visible
          if constexpr (NTAPS >= 7) {     // (live-range seam for the allocator, as at the hand-over to the last body)
#pragma unroll
            for (int g = 0; g < 8; ++g) asm volatile("" : "+v"(acc[g][0]), "+v"(acc[g][1]));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      v2q_for(phase, std::make_integer_sequence<int, Q::NPH>());
    };
    for (int cb = cb_begin; cb + 4 < cb_end; cb += 4) quad(cb, std::false_type());
#pragma unroll
    for (int g = 0; g < 8; ++g) asm volatile("" : "+v"(acc[g][0]), "+v"(acc[g][1]));
    quad(cb_end - 4, std::true_type());
#pragma unroll
    for (int g = 0; g < 8; ++g) asm volatile("" : "+v"(acc[g][0]), "+v"(acc[g][1]));     // (and again towards the epilogue)
  }
  // (nothing is in flight here: the last body waited for its last set with vmcnt(0) and its end-of-block barrier has been passed
  // by every wave, so the slab buffers may become the epilogue's scratch)
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  if (NTAPS != 3 && slice) {               // (the three-tap ResNet form has no K-split tail)
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
  if constexpr (NTAPS == 3) {
    // the bin's output rows: rowmap[m] + bin, as pointer offsets (taken here, behind the loop, so that nothing derived from them
    // lives through it)
    GemmArgs q = p;
    const int b = __builtin_amdgcn_readfirstlane(bin);
    if (p.Ysb) q.Ysb = reinterpret_cast<char*>(p.Ysb) + (int64_t)b * p.ldsb * 4;
    if (p.Y) q.Y = p.Y + (int64_t)b * p.ldy;
    if (p.R) q.R = p.R + (int64_t)b * p.ldr;
    if constexpr (RES) store_wave_tile_n32_res(q, acc, m0, n0 + wave * 32, lane_e, wave, smem);
    else store_wave_tile_n32_f6(q, acc, m0, n0 + wave * 32, lane_e, wave, smem);
  } else {
    if constexpr (OUT_F6) store_wave_tile_n32_f6(p, acc, m0, n0 + wave * 32, lane_e, wave, smem);
    else store_wave_tile_n32<64, true>(p, acc, m0, n0 + wave * 32, lane_e, wave, smem);
  }
}

// Workgroups [0, nMt * nNt) are whole tiles; with S > 0 the grid continues with the K-split slices of the tail tiles exactly as in
// gemm_bf16x3_w14p2_kernel (raw partial sums to GemmArgs::partial, finished by f6v2_tail_reduce_kernel).
// GemmArgs::nbin > 1 (the ResNet form): an "M tile" is a pair (128 time rows, frequency bin); bin b reads its windows b * bin_x_bytes
// further into the rows of A and writes output rows rowmap[m] + b (no K-split tail in this form).
template <int NTAPS, bool OUT_F6, bool RES = false>
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
  int bin = 0;
  if (NTAPS == 3 && p.nbin > 1) {
    bin = mt % p.nbin;
    mt /= p.nbin;
  }
  f6v2_tile<NTAPS, OUT_F6, RES>(p, mt * V2_BM, nt * V2_BN, smem_v2, cb_begin, cb_end, slice, bin);
}

// One thread per (tail row, 4 channels), the eight lanes of a 32-channel block side by side: ordered sum of the K slices, BN scale /
// shift + activation, then fp32 and / or the split-blocked row or the two-unit block (what the layer's whole tiles write through
// their epilogues).  The block's two maxima are taken over its eight lanes, every lane codes its four values into 24 bits, and
// lanes 0-5 assemble dword j of the 192-bit code string from two neighbours' pieces.  (One thread per block, everything serial,
// took 14-16 us per layer for 1.4 % of its work.)  Every branch below is uniform over a group of eight lanes.
template <int S>
__global__ __launch_bounds__(256) void f6v2_tail_reduce_kernel(GemmArgs p, int mt0) {
  const int quads = p.Npad >> 2;            // Npad % 32 == 0: a group of eight lanes never straddles two rows
  const int64_t rows = (int64_t)p.tail_mt * V2_BM;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * quads) return;
  const int64_t r = i / quads;
  const int n = (int)(i - r * quads) * 4;
  const int64_t m = (int64_t)mt0 * V2_BM + r;
  if (m >= p.M) return;
  bool zero;
  const int orow = out_row(p, (int)m, zero);
  if (orow < 0) return;
  // S is a template parameter: the S loads of a thread are issued together; the slices are still added in ascending order
  f32x4 t[S];
#pragma unroll
  for (int s = 0; s < S; ++s) t[s] = *reinterpret_cast<const f32x4*>(p.partial + ((int64_t)s * rows + r) * p.Npad + n);
  f32x4 acc = t[0];
#pragma unroll
  for (int s = 1; s < S; ++s) acc += t[s];
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (n + e < p.N && !zero) v[e] = apply_act(fmaf(acc[e], p.scale[n + e], p.shift[n + e]), p.act, p.alpha ? p.alpha[n + e] : 0.f);
  if (p.Y && n < p.N) *reinterpret_cast<f32x4*>(p.Y + (int64_t)orow * p.ldy + n) = v;     // N % 4 == 0
  if (!p.Ysb || n >= p.ldsb) return;
  char* dst = reinterpret_cast<char*>(p.Ysb) + (int64_t)orow * p.ldsb * 4 + (n >> 5) * 128;
  const int j = (n & 31) >> 2;              // this lane's place among the eight of its block
  v *= p.sb_mul;
  uint32_t h01, l01, h23, l23;
  split2t<true>(v[0], v[1], h01, l01);
  split2t<true>(v[2], v[3], h23, l23);
  ovf_report(p.ovf, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
  *reinterpret_cast<uint2*>(dst + 8 * j) = make_uint2(h01, h23);
  if (!p.ysb_f6) {
    *reinterpret_cast<uint2*>(dst + 64 + 8 * j) = make_uint2(l01, l23);
    return;
  }
  const f16x2_t ha = __builtin_bit_cast(f16x2_t, h01), hb = __builtin_bit_cast(f16x2_t, h23);
  const float hf[4] = {(float)ha[0], (float)ha[1], (float)hb[0], (float)hb[1]};
  float lf[4], mh = 0.f, ml = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    lf[e] = v[e] - hf[e];
    mh = fmaxf(mh, fabsf(hf[e]));
    ml = fmaxf(ml, fabsf(lf[e]));
  }
#pragma unroll
  for (int d = 1; d < 8; d <<= 1) {
    mh = fmaxf(mh, __shfl_xor(mh, d, 8));
    ml = fmaxf(ml, __shfl_xor(ml, d, 8));
  }
  uint32_t bh, bl;
  const float ih = e8m0_of(mh, bh), il = e8m0_of(ml, bl);
  uint32_t ph = 0, pl = 0;                  // codes 4 j .. 4 j + 3 = bits 24 j .. 24 j + 23 of the block's code string
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ph |= e2m3_code(hf[e], ih) << (6 * e);
    pl |= e2m3_code(lf[e], il) << (6 * e);
  }
  // dword j of the string (j < 6) = bits 32 j .. 32 j + 31: piece a = 32 j / 24 from bit 32 j % 24 on, then piece a + 1
  const int a = (4 * j) / 3, off = 8 * (j % 3);
  const uint32_t ha0 = __shfl(ph, a & 7, 8), ha1 = __shfl(ph, (a + 1) & 7, 8);
  const uint32_t la0 = __shfl(pl, a & 7, 8), la1 = __shfl(pl, (a + 1) & 7, 8);
  const uint32_t sc2 = bh | (bl << 8);
  const uint32_t wh = j < 6 ? (ha0 >> off) | (ha1 << (24 - off)) : (j == 6 ? sc2 : 0u);
  const uint32_t wl = j < 6 ? (la0 >> off) | (la1 << (24 - off)) : (j == 6 ? sc2 : 0u);
  // chunk 4 / 5: code dwords 0-3 of hi / lo; chunk 6 / 7: {code dwords 4, 5 | scale dword | pad}
  const int oh = j < 4 ? 64 + 4 * j : 96 + 4 * (j - 4);
  *reinterpret_cast<uint32_t*>(dst + oh) = wh;
  *reinterpret_cast<uint32_t*>(dst + oh + 16) = wl;
  if (m == p.M - 1) {                      // eight zero rows behind the value's last row (store_wave_tile_n32_f6)
    for (int k = 1; k <= 8; ++k) *reinterpret_cast<uint4*>(dst + (int64_t)k * p.ldsb * 4 + 16 * j) = uint4{0u, 0u, 0u, 0u};
  }
}


#undef V2_GLD16

// a.Xsb = activations in the two-unit block format (row stride a.ldsbx channels), a.Wfr / a.Wx6 = main / cross weights
// (xvec_api.hip, upload_layer), a.K = taps * a.cin, taps 5, 7 or 9, a.cin % 32 == 0.  a.tail_mt / a.ksplit / a.partial: the K-split tail
// (gemm_bf16x3_tail_plan), decided at plan time.
hipError_t launch_gemm_f16f6(const GemmArgs& a, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  const int taps = a.cin > 0 ? a.K / a.cin : 0;
  // Only the widths the reference's graphs contain are instantiated (tdnn: 5, 5, 7; extended tdnn: 5, 5, 7, 9).
  const bool bins = a.nbin > 1;             // the ResNet form: three taps along time, the rows wider than the window (ldsbx > cin)
  if ((taps != 3 && taps != 5 && taps != 7 && taps != 9) || (a.cin & 31) || (bins ? a.ldsbx < a.cin || taps != 3 : a.ldsbx != a.cin || taps == 3) ||
      !a.Wx6 || !a.Wfr || a.a_pitch || a.pool_part || (a.R && !bins) || (a.N & 3))
    return hipErrorInvalidValue;
  static std::mutex mu;
  static bool attr_set[64] = {};
  const size_t smem = (size_t)kV2Ring * kV2SlabBytes;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (!attr_set[dev & 63]) {
      const void* ks[] = {reinterpret_cast<const void*>(gemm_f6v2_kernel<5, false>), reinterpret_cast<const void*>(gemm_f6v2_kernel<7, false>),
                          reinterpret_cast<const void*>(gemm_f6v2_kernel<5, true>), reinterpret_cast<const void*>(gemm_f6v2_kernel<7, true>),
                          reinterpret_cast<const void*>(gemm_f6v2_kernel<9, false>), reinterpret_cast<const void*>(gemm_f6v2_kernel<9, true>),
                          reinterpret_cast<const void*>(gemm_f6v2_kernel<3, true>), reinterpret_cast<const void*>(gemm_f6v2_kernel<3, true, true>)};
      for (const void* k : ks) {
        const hipError_t r = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kV2Ring * kV2SlabBytes);
        if (r != hipSuccess) return r;
      }
      attr_set[dev & 63] = true;
    }
  }
  const int nMt = ((a.M + V2_BM - 1) / V2_BM) * (bins ? a.nbin : 1), nNt = a.Npad / V2_BN, ncb = a.cin >> 5;
  if (ncb & 3) return hipErrorInvalidValue;      // the bodies take channel blocks in quads
  const bool tail = !bins && a.tail_mt > 0 && (a.ksplit == 2 || a.ksplit == 4 || a.ksplit == 8) && a.partial && a.tail_mt < nMt && ncb % a.ksplit == 0 &&
                    (ncb / a.ksplit) % 4 == 0;
  const int nMain = tail ? nMt - a.tail_mt : nMt;
  const int S = tail ? a.ksplit : 0;
  const dim3 grid(nMain * nNt + (tail ? a.tail_mt * nNt * a.ksplit : 0)), block(256);
  const bool out_f6 = a.ysb_f6 && a.Ysb && (bins || !a.Y);
  if (a.ysb_f6 && !out_f6) return hipErrorInvalidValue;        // (TDNN forms: the block format is written only when it is the layer's one output)
  if (taps == 3) {                       // <3, true, true>: the residual epilogue, block or split-blocked output by GemmArgs::ysb_f6
    if (out_f6 && !a.R && !a.Y) hipLaunchKernelGGL((gemm_f6v2_kernel<3, true>), grid, block, smem, s, a, nMain, nNt, S);
    else                        hipLaunchKernelGGL((gemm_f6v2_kernel<3, true, true>), grid, block, smem, s, a, nMain, nNt, S);
  } else if (taps == 5) {
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
  const int64_t total = (int64_t)a.tail_mt * V2_BM * (a.Npad >> 2);
  const dim3 rgrid((unsigned)((total + 255) / 256)), rblock(256);
  switch (a.ksplit) {
    case 2: hipLaunchKernelGGL(f6v2_tail_reduce_kernel<2>, rgrid, rblock, 0, s, a, nMain); break;
    case 4: hipLaunchKernelGGL(f6v2_tail_reduce_kernel<4>, rgrid, rblock, 0, s, a, nMain); break;
    case 8: hipLaunchKernelGGL(f6v2_tail_reduce_kernel<8>, rgrid, rblock, 0, s, a, nMain); break;
    default: return hipErrorInvalidValue;             // gemm_bf16x3_tail_plan picks 2, 4 or 8 slices
  }
  return hipGetLastError();
}

}  // namespace xv
