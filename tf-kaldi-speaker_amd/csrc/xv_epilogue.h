// Shared GEMM epilogues (gfx950): the 16x16 accumulator layout of the split-precision product kernels
// (store_wave_tile_n32*), the 32x32 layout of the fp32 kernel and the lab kernels (store_wave_tile), and the
// "split-blocked" (SB) activation format of the bf16x3 / f16x3 path.
//
// SB format: a row of C channels is stored as ceil(C/32) blocks of 128 bytes; block j holds
// channels 32j..32j+31 as [32 x bf16 hi | 32 x bf16 lo] with hi = rn_bf16(x), lo = rn_bf16(x - hi)
// (hi + lo carries 16 significant bits).  Same bytes per element as fp32; one (row, block) is
// one full 128-byte line, and 16-byte chunk q of a block is plane q>>2 (hi / lo), k chunk q&3 (= lane>>4) of the
// v_mfma_f32_16x16x32_bf16 A/B fragments.
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

// debug (tools/gemm_trace.py, -DXV_GEMM_TRACE builds only): phase stamps 4..7 of the workgroup's first thread
#ifdef XV_GEMM_TRACE
#define XV_EPI_STAMP(p, i)                                                                                          \
  do {                                                                                                              \
    if ((p).trace && threadIdx.x == 0) (p).trace[(int64_t)blockIdx.x * 8 + (i)] = (long long)__builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define XV_EPI_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define XV_EPI_STAMP(p, i) do {} while (0)
#define XV_EPI_DRAIN() do {} while (0)
#endif

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// hi/lo split of two fp32 values, packed {a | b << 16}.
//  bf16 (F16 = false): v_cvt_pk_bf16_f32 (RNE, NaN-safe), shl/and, v_pk_add_f32, v_cvt_pk_bf16_f32 = 2.5 VALU per value;
//        hi + lo carries 16 significand bits over the full fp32 range.
//  fp16 (F16 = true):  hi = rn_f16(x), lo = rn_f16(x - hi): 22 significand bits for |x| in [2^-3, 65504]; below that the
//        low half goes subnormal (absolute error <= 2^-25), above it the value overflows to inf (the host checks).
template <bool F16>
__device__ __forceinline__ void split2t(float a, float b, uint32_t& hi, uint32_t& lo) {
  const f32x2_t v = {a, b};
  if constexpr (F16) {
    const f16x2_t hh = __builtin_convertvector(v, f16x2_t);
    hi = __builtin_bit_cast(uint32_t, hh);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(v - __builtin_convertvector(hh, f32x2_t), f16x2_t));
  } else {
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
    const f32x2_t hf = {__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(v - hf, bf16x2_t));
  }
}
// fp16 range guard: callers fold every value they convert into `m` (one v_max3_f32 per pair) and report once per wave
constexpr float kF16Max = 65504.f;
__device__ __forceinline__ void ovf_track(float& m, float a, float b) { m = fmaxf(m, fmaxf(fabsf(a), fabsf(b))); }
__device__ __forceinline__ void ovf_report(int* flag, float m) {
  if (flag && !(m <= kF16Max)) *flag = 1;              // also true for NaN; every reporter writes the same value
}
// the same guard on the converted halves (hot epilogue: integer ops on registers that exist anyway, no extra live floats):
// a half is inf / NaN iff its exponent field is all ones, i.e. adding 1 to the field carries into the sign bit
__device__ __forceinline__ void ovf_bits(uint32_t& bad, uint32_t hi) { bad |= (hi & 0x7C007C00u) + 0x04000400u; }
__device__ __forceinline__ void ovf_report_bits(int* flag, uint32_t bad) {
  if (flag && (bad & 0x80008000u)) *flag = 1;
}
// run-time form for the kernels off the hot path (f16 is uniform over the launch)
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo, int f16) {
  if (f16) split2t<true>(a, b, hi, lo); else split2t<false>(a, b, hi, lo);
}

// one product tile of the split GEMM: v_mfma_f32_32x32x16_bf16 or _f16 (same rate, same fragment layout)
template <bool F16>
__device__ __forceinline__ f32x16 mfma_split(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// The product kernels use the 16x16x32 shape: same FLOP per cycle as 32x32x16, but the chip holds a visibly higher clock
// on it under this load (MI355X_MICROARCH.md "DVFS give-back" item 7; measured in this kernel: profiles/README.md).
template <bool F16>
__device__ __forceinline__ f32x4 mfma_split16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// XCD-aware bijective remap of a 1-D grid (ids congruent mod 8 share an XCD and its L2).
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7, i = id >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// ---------------------------------------------------------------------------------------------
// Accumulator orientation used by every GEMM kernel here: the WEIGHT fragment is the MFMA A
// operand and the ACTIVATION fragment the B operand, so a 32x32 result tile is D[n][m]:
//   lane & 31            -> frame m (row of the output matrix)
//   (e&3) + 8*(e>>2) + 4*(lane>>5) -> channel n, i.e. registers 4q..4q+3 are FOUR CONSECUTIVE
//                                     channels 8q + 4*(lane>>5) .. +3 of one frame.
// A lane can therefore apply per-channel scale/shift with float4 loads, split to bf16 in packed
// pairs and emit 8/16-byte LDS writes with no cross-lane traffic.

// Output row of GEMM row m.  rowmap values: r >= 0 store at row r; -1 skip; r <= -2 store ZEROS at row -r - 2 (a border
// position of a zero-bordered grid value that this GEMM row happens to cover: csrc/grid.hip -- this is what keeps the
// border zero without a memset of the whole output per layer).
__device__ __forceinline__ int out_row(const GemmArgs& p, int m, bool& zero) {
  zero = false;
  if (m >= p.M) return -1;
  int r = p.rowmap ? p.rowmap[m] : m;
  if (r < -1) { zero = true; r = -r - 2; }
  return r;
}

// Scalar fallback (any N / alignment): one element at a time, fp32 and/or split-blocked.
__device__ __forceinline__ void store_tile_scalar(const GemmArgs& p, const f32x16& acc, int mbase, int nbase, int lane) {
  const int r32 = lane & 31, h = lane >> 5;
  const int m = mbase + r32;
  bool zero;
  const int orow = out_row(p, m, zero);
  if (orow < 0) return;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int n = nbase + (e & 3) + 8 * (e >> 2) + 4 * h;
    const bool nok = n < p.N;
    float v = 0.f;
    if (nok && !zero) {
      v = p.raw ? acc[e] : fmaf(acc[e], p.scale[n], p.shift[n]);
      if (p.R) v += p.R[(int64_t)orow * p.ldr + n];
      v = apply_act(v, p.raw ? ACT_NONE : p.act, p.alpha ? p.alpha[n] : 0.f);
    }
    if (p.Y && nok) p.Y[(int64_t)orow * p.ldy + n] = v;
    if (p.Ysb && n < p.ldsb) {
      uint32_t hi, lo;
      const float vs = v * p.sb_mul;
      split2(vs, 0.f, hi, lo, p.f16);
      if (p.f16 && !(fabsf(vs) <= kF16Max) && p.ovf) *p.ovf = 1;
      uint16_t* blk = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(p.Ysb) + (int64_t)orow * p.ldsb * 4 + (n >> 5) * 128);
      blk[n & 31] = (uint16_t)hi;
      blk[32 + (n & 31)] = (uint16_t)lo;
    }
  }
}

// The same for one 16x16 accumulator tile of the split kernels (lane & 15 -> frame, 4 * (lane >> 4) + i -> channel).
__device__ __forceinline__ void store_tile16_scalar(const GemmArgs& p, const f32x4& acc, int mbase, int nbase, int lane) {
  const int m = mbase + (lane & 15);
  bool zero;
  const int orow = out_row(p, m, zero);
  if (orow < 0) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int n = nbase + 4 * (lane >> 4) + e;
    const bool nok = n < p.N;
    float v = 0.f;
    if (nok && !zero) {
      v = p.raw ? acc[e] : fmaf(acc[e], p.scale[n], p.shift[n]);
      if (p.R) v += p.R[(int64_t)orow * p.ldr + n];
      v = apply_act(v, p.raw ? ACT_NONE : p.act, p.alpha ? p.alpha[n] : 0.f);
    }
    if (p.Y && nok) p.Y[(int64_t)orow * p.ldy + n] = v;
    if (p.Ysb && n < p.ldsb) {
      uint32_t hi, lo;
      const float vs = v * p.sb_mul;
      split2(vs, 0.f, hi, lo, p.f16);
      if (p.f16 && !(fabsf(vs) <= kF16Max) && p.ovf) *p.ovf = 1;
      uint16_t* blk = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(p.Ysb) + (int64_t)orow * p.ldsb * 4 + (n >> 5) * 128);
      blk[n & 31] = (uint16_t)hi;
      blk[32 + (n & 31)] = (uint16_t)lo;
    }
  }
}

// Chunk swizzle of the 128-byte scratch rows of the 128 x 32 wave-tile epilogues: (row & 7) ^ ((row >> 3) & 1).  LDS stores see 32
// banks (one 128-byte row), so the eight consecutive rows of a ds_write_b128 lane group need eight different chunks: row & 7;
// loads see 64 banks (two rows) and serve a ds_read_b128 in groups of lanes {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}: with
// one row per lane (xv_f6.h) rows r and r + 24 / r + 12 of one parity would collide under row & 7 alone, bit 3 of the row
// separates them.  Measured (SQ_LDS_BANK_CONFLICT per 64-row pass): row & 7 alone 0 / 128 / 64 for the pooling / fp6 / split-
// blocked forms, (row >> 1) & 7 64 / 128 / 64; the split-blocked form's 64 are its ds_write_b64 (16 rows, one 8-byte half: 2-way
// by construction).  tests/analysis/lds_bank_model.py.
__device__ __forceinline__ int swz8(int row) { return (row & 7) ^ ((row >> 3) & 1); }

// Ordering point between a wave's own LDS writes and its cross-lane read-back.  The scratch is private to the
// wave and DS instructions of one wave execute in order, so no workgroup barrier (and no s_barrier at all) is
// needed -- only that the compiler keeps the program order.
__device__ __forceinline__ void wave_lds_sync() {
  // No fence builtin here: on gfx950 vmcnt counts loads AND stores, and a release / acquire pair makes the compiler
  // drain it -- the wave would sit until the global stores of the previous pass are acknowledged (microseconds under
  // load) before it may restage the scratch.  Only compiler order (the asm memory clobber) and LDS order are needed.
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// LDS-staged epilogue of a 64x64 wave tile acc[ni][mi] (2x2 MFMA tiles, orientation above).
// The wave writes its post-activation tile into a private LDS scratch shaped like the
// destination rows (256 bytes per frame: two SB blocks, or 64 floats; 16-byte chunks XOR-swizzled
// by frame & 15), then reads it back 16 bytes per lane and stores whole 256-byte row segments.
// ROWS = frames staged per pass: 64 (16 KB of scratch per wave, one pass; required by the fused pooling)
// or 32 (8 KB per wave, two passes -- lets three workgroups share a CU's LDS).
// Requirements (wide_epilogue_ok): N % 4 == 0, ldy % 4 == 0, 16-byte aligned Y.  `lds` is the workgroup's
// LDS base, ROWS * 256 bytes per wave, no longer read by the main loop (the caller's last K-loop barrier
// guarantees that); there is no workgroup barrier inside.
// MI = 32-frame MFMA tiles per wave along M (2: 64-frame wave tile; 1: the half-height tail tiles).
template <int ACT, int ROWS, int MI>
__device__ __forceinline__ void store_wave_tile_impl(const GemmArgs& p, const f32x16 (&acc)[2][MI], int mbase, int nbase,
                                                     int lane, int wave, char* lds) {
  static_assert((ROWS == 64 || ROWS == 32) && ROWS <= 32 * MI, "ROWS");
  constexpr int NPASS = 32 * MI / ROWS, MIP = ROWS / 32;      // passes, MFMA row tiles per pass
  const int r32 = lane & 31, h = lane >> 5;
  char* scratch = lds + wave * (ROWS * 256);
  const int rrow = lane >> 4, rchunk = lane & 15;       // read-back map: 4 frames x 16 chunks per pass

  auto value4 = [&](const f32x16& t, int q, const f32x4& sc, const f32x4& sh, const f32x4& al) -> f32x4 {
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = apply_act(fmaf(t[4 * q + i], sc[i], sh[i]), ACT < 0 ? p.act : ACT, al[i]);
    return v;
  };
  auto params = [&](int n4, f32x4& sc, f32x4& sh, f32x4& al) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (p.raw) {                                    // split-K partial: accumulators unchanged
      sc = f32x4{1.f, 1.f, 1.f, 1.f};
      sh = z;
      al = z;
      return;
    }
    const bool ok = n4 < p.N;
    sc = ok ? *reinterpret_cast<const f32x4*>(p.scale + n4) : z;
    sh = ok ? *reinterpret_cast<const f32x4*>(p.shift + n4) : z;
    al = (ok && p.alpha) ? *reinterpret_cast<const f32x4*>(p.alpha + n4) : z;
  };
  // stage fp32 values of pass `ps` (post-activation unless `pre_act`): chunk ni*8 + 2q + h of frame row
  auto stage_f32 = [&](int ps, bool pre_act) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 sc, sh, al;
        params(nbase + ni * 32 + 8 * q + 4 * h, sc, sh, al);
#pragma unroll
        for (int ml = 0; ml < MIP; ++ml) {
          const int row = ml * 32 + r32;
          const f32x16& t = acc[ni][ps * MIP + ml];
          f32x4 v;
          if (pre_act) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaf(t[4 * q + i], sc[i], sh[i]);
          } else {
            v = value4(t, q, sc, sh, al);
          }
          *reinterpret_cast<f32x4*>(scratch + row * 256 + (((ni * 8 + 2 * q + h) ^ (row & 15)) << 4)) = v;
        }
      }
  };

  if (p.R) {
    // residual form (model/resnet.py:84-85,147-148): y = act(bn(conv) + shortcut).  Stage the BN output in
    // LDS, then add the fp32 shortcut row-wise (coalesced 16-byte loads), activate, and store fp32 / SB.
    const int n = nbase + rchunk * 4;
    const bool nok = n < p.N;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 al4 = (nok && p.alpha) ? *reinterpret_cast<const f32x4*>(p.alpha + n) : z;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      stage_f32(ps, true);
      wave_lds_sync();
#pragma unroll 4
      for (int it = 0; it < ROWS / 4; ++it) {
        const int row = it * 4 + rrow;
        const int m = mbase + ps * ROWS + row;
        f32x4 v = *reinterpret_cast<const f32x4*>(scratch + row * 256 + ((rchunk ^ (row & 15)) << 4));
        bool zero;
        const int orow = out_row(p, m, zero);
        if (orow < 0) continue;
        if (nok) {
          if (zero) {
            v = z;
          } else {
            v += *reinterpret_cast<const f32x4*>(p.R + (int64_t)orow * p.ldr + n);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i], p.act, al4[i]);
          }
          if (p.Y) *reinterpret_cast<f32x4*>(p.Y + (int64_t)orow * p.ldy + n) = v;
        } else {
          v = z;
        }
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
      wave_lds_sync();
    }
    return;
  }
  if (p.Ysb) {
    const bool blk_ok = nbase + (rchunk >> 3) * 32 < p.ldsb;     // SB block exists in the output row
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 sc, sh, al;
          params(nbase + ni * 32 + 8 * q + 4 * h, sc, sh, al);     // padding channels: scale = shift = 0 -> 0
#pragma unroll
          for (int ml = 0; ml < MIP; ++ml) {
            const f32x4 v = value4(acc[ni][ps * MIP + ml], q, sc, sh, al) * p.sb_mul;
            uint32_t h01, l01, h23, l23;
            split2(v[0], v[1], h01, l01, p.f16);
            split2(v[2], v[3], h23, l23, p.f16);
            if (p.f16) ovf_report(p.ovf, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
            const int row = ml * 32 + r32;
            char* rp = scratch + row * 256 + 8 * h;
            *reinterpret_cast<uint2*>(rp + (((ni * 8 + q) ^ (row & 15)) << 4)) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(rp + (((ni * 8 + 4 + q) ^ (row & 15)) << 4)) = make_uint2(l01, l23);
          }
        }
      wave_lds_sync();
#pragma unroll 4
      for (int it = 0; it < ROWS / 4; ++it) {
        const int row = it * 4 + rrow;
        const int m = mbase + ps * ROWS + row;
        f32x4 v = *reinterpret_cast<const f32x4*>(scratch + row * 256 + ((rchunk ^ (row & 15)) << 4));
        bool zero;
        const int orow = out_row(p, m, zero);
        if (zero) v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (orow >= 0 && blk_ok)
          *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(p.Ysb) + (int64_t)orow * p.ldsb * 4 + (nbase >> 5) * 128 +
                                    rchunk * 16) = v;
      }
      wave_lds_sync();
    }
  }
  if (p.Y || p.pool_part) {
    const int n = nbase + rchunk * 4;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      stage_f32(ps, false);
      wave_lds_sync();
      if (p.Y) {
#pragma unroll 4
        for (int it = 0; it < ROWS / 4; ++it) {
          const int row = it * 4 + rrow;
          const int m = mbase + ps * ROWS + row;
          f32x4 v = *reinterpret_cast<const f32x4*>(scratch + row * 256 + ((rchunk ^ (row & 15)) << 4));
          bool zero;
          const int orow = out_row(p, m, zero);
          if (zero) v = f32x4{0.f, 0.f, 0.f, 0.f};
          if (orow >= 0 && n < p.N) *reinterpret_cast<f32x4*>(p.Y + (int64_t)orow * p.ldy + n) = v;
        }
      }
      if (NPASS > 1) wave_lds_sync();
    }
  }
  if (ROWS == 64 && MI == 2 && p.pool_part) {
    // fused statistics pooling: lane = channel; for every utterance segment inside the 64-frame
    // tile store (sum x, sum (x - segment mean)^2) -- two passes over the LDS copy, so the merge in
    // pool_finalize_kernel (Chan et al.) is as accurate as the reference's two-pass variance.
    // Lane map: cq = lane & 15 -> columns 4cq..4cq+3 (one 16-byte LDS chunk), rg = lane >> 4 -> frames
    // t == rg (mod 4); the four frame groups are combined with two xor-shuffles.
    const int cq = lane & 15, rg = lane >> 4;
    const int n = nbase + 4 * cq;
    const bool nok = n < p.N;
    const int my_utt = (mbase + lane < p.M) ? p.pool_row2utt[mbase + lane] : -1;   // lane r: utterance of frame r
    const int tile64 = mbase >> 6;
    auto row4 = [&](int t) -> f32x4 {
      return *reinterpret_cast<const f32x4*>(scratch + t * 256 + ((cq ^ (t & 15)) << 4));
    };
    auto groups_sum = [&](f32x4 v) -> f32x4 {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] += __shfl_xor(v[i], 16, 64);
        v[i] += __shfl_xor(v[i], 32, 64);
      }
      return v;
    };
    int r = 0;
    while (r < 64) {
      const int b = __builtin_amdgcn_readlane(my_utt, r);
      int re = r + 1;
      while (re < 64 && __builtin_amdgcn_readlane(my_utt, re) == b) ++re;
      if (b >= 0) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        f32x4 s1 = z;
        for (int t0 = r & ~3; t0 < re; t0 += 4) {
          const int t = t0 + rg;
          if (t >= r && t < re) s1 += row4(t);
        }
        s1 = groups_sum(s1);
        const f32x4 mu = s1 / (float)(re - r);
        f32x4 m2 = z;
        for (int t0 = r & ~3; t0 < re; t0 += 4) {
          const int t = t0 + rg;
          if (t >= r && t < re) {
            const f32x4 d = row4(t) - mu;
            m2 += d * d;
          }
        }
        m2 = groups_sum(m2);
        if (nok && rg == 0) {
          const int64_t slot = (int64_t)p.pool_slotbase[b] + tile64;
          *reinterpret_cast<f32x4*>(p.pool_part + (slot * 2) * p.N + n) = s1;
          *reinterpret_cast<f32x4*>(p.pool_part + (slot * 2 + 1) * p.N + n) = m2;
        }
      }
      r = re;
    }
  }
}

// LDS-staged epilogue of a 128-frame x 32-channel wave tile acc[ft][ct] (8 x 2 accumulator tiles of 16 frames x 16
// channels, v_mfma_f32_16x16x32: lane & 15 -> frame, registers -> channels 4 * (lane >> 4) .. + 3; the "one wave per
// 32-channel block" layout of the split kernels).  A row of the wave tile is exactly one SB block
// (128 bytes) or 32 floats, so the scratch rows are 128 bytes (8 chunks, XOR-swizzled by swz8(frame), below) and the
// read-back hands 8 lanes one whole 128-byte line.  ROWS frames per pass (ROWS * 128 bytes of scratch per wave);
// the fused statistics pooling needs ROWS == 64 (its partial slots are per 64-frame tile).
template <int ACT, int ROWS, bool F16>
__device__ __forceinline__ void store_wave_tile_n32_impl(const GemmArgs& p, const f32x4 (&acc)[8][2], int mbase, int nbase,
                                                         int lane, int wave, char* lds) {
  static_assert(ROWS == 64, "ROWS");
  constexpr int NPASS = 128 / ROWS, FPP = ROWS / 16;     // passes, 16-frame accumulator tiles per pass
  const int c16 = lane & 15, g4 = lane >> 4;
  char* scratch = lds + wave * (ROWS * 128);
  const int rrow = lane >> 3, rchunk = lane & 7;        // read-back map: 8 frames x 8 chunks per pass
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  uint32_t bad = 0;                                     // F16 range guard: exponent-overflow bits of the converted hi halves

  // GemmArgs::sb_mul (the power of two the split-blocked copy is kept at): when that copy is the only output it is folded into
  // scale / shift here (every activation but tanh is positively homogeneous, and tanh layers have sb_mul == 1); otherwise the
  // values are multiplied right before their split (uniform branch)
  const bool fold = p.Ysb && !p.Y && !p.R && !p.pool_part && !p.raw;
  const float sbm = fold ? 1.f : p.sb_mul;
  f32x4 sc[2], sh[2];                                   // this lane's channels 16 ct + 4 g4 .. +3, ct = 0, 1
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int n4 = nbase + 16 * ct + 4 * g4;
    if (p.raw) {
      sc[ct] = f32x4{1.f, 1.f, 1.f, 1.f};
      sh[ct] = z;
    } else {
      const bool ok = n4 < p.N;
      sc[ct] = ok ? *reinterpret_cast<const f32x4*>(p.scale + n4) : z;
      sh[ct] = ok ? *reinterpret_cast<const f32x4*>(p.shift + n4) : z;
      if (fold) { sc[ct] *= p.sb_mul; sh[ct] *= p.sb_mul; }
    }
  }
  // Row map entries of every row this lane will store (8 per pass), loaded BEFORE the first store: vmcnt counts loads and
  // stores together and in order, so a row-map load issued behind the stores of the previous pass could only be consumed
  // once those stores are acknowledged (the convolutions' epilogue spent most of its time there).
  int rmap[NPASS][ROWS / 8];
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
    for (int it = 0; it < ROWS / 8; ++it) {
      const int m = mbase + ps * ROWS + it * 8 + rrow;
      rmap[ps][it] = m < p.M ? (p.rowmap ? p.rowmap[m] : m) : -1;
    }
  auto out_row_pre = [&](int ps, int it, bool& zero) -> int {      // out_row() on the preloaded entry
    int r = rmap[ps][it];
    zero = r < -1;
    return zero ? -r - 2 : r;
  };
  XV_EPI_DRAIN();                                        // (trace builds: make the parameter-load latency visible)
  XV_EPI_STAMP(p, 4);
  auto value4 = [&](const f32x4& t, int ct, bool pre_act) -> f32x4 {
    f32x4 al = z;                                        // PReLU slopes: loaded at the point of use (registers are scarce)
    if (ACT < 0 && !pre_act && p.alpha && !p.raw && nbase + 16 * ct + 4 * g4 < p.N)
      al = *reinterpret_cast<const f32x4*>(p.alpha + nbase + 16 * ct + 4 * g4);
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float y = fmaf(t[i], sc[ct][i], sh[ct][i]);
      v[i] = pre_act ? y : apply_act(y, ACT < 0 ? p.act : ACT, al[i]);
    }
    return v;
  };
  auto stage_f32 = [&](int ps, bool pre_act) {          // fp32 chunk of channels 16 ct + 4 g4: index 4 ct + g4
#pragma unroll
    for (int fl = 0; fl < FPP; ++fl)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int row = fl * 16 + c16;
        *reinterpret_cast<f32x4*>(scratch + row * 128 + (((4 * ct + g4) ^ swz8(row)) << 4)) = value4(acc[ps * FPP + fl][ct], ct, pre_act);
      }
  };
  const int n = nbase + rchunk * 4;                      // read-back channels of this lane (fp32 forms)

  if (p.R) {
    const bool nok = n < p.N;
    const f32x4 al4 = (nok && p.alpha) ? *reinterpret_cast<const f32x4*>(p.alpha + n) : z;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      stage_f32(ps, true);
      wave_lds_sync();
#pragma unroll 2
      for (int it = 0; it < ROWS / 8; ++it) {
        const int row = it * 8 + rrow;
        f32x4 v = *reinterpret_cast<const f32x4*>(scratch + row * 128 + ((rchunk ^ swz8(row)) << 4));
        bool zero;
        const int orow = out_row_pre(ps, it, zero);
        if (orow < 0) continue;
        if (nok) {
          if (zero) {
            v = z;
          } else {
            v += *reinterpret_cast<const f32x4*>(p.R + (int64_t)orow * p.ldr + n);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i], p.act, al4[i]);
          }
          if (p.Y) *reinterpret_cast<f32x4*>(p.Y + (int64_t)orow * p.ldy + n) = v;
        } else {
          v = z;
        }
        if (p.Ysb && n < p.ldsb) {
          uint32_t h01, l01, h23, l23;
          v *= p.sb_mul;
          split2t<F16>(v[0], v[1], h01, l01);
          split2t<F16>(v[2], v[3], h23, l23);
          if constexpr (F16) { ovf_bits(bad, h01); ovf_bits(bad, h23); }
          char* blk = reinterpret_cast<char*>(p.Ysb) + (int64_t)orow * p.ldsb * 4 + (n >> 5) * 128 + (n & 31) * 2;
          *reinterpret_cast<uint2*>(blk) = make_uint2(h01, h23);
          *reinterpret_cast<uint2*>(blk + 64) = make_uint2(l01, l23);
        }
      }
      wave_lds_sync();
    }
    if constexpr (F16) ovf_report_bits(p.ovf, bad);
    return;
  }
  if (p.Ysb) {
    const bool blk_ok = nbase < p.ldsb;                   // the SB block exists in the output row
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
      for (int fl = 0; fl < FPP; ++fl)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          f32x4 v = value4(acc[ps * FPP + fl][ct], ct, false);   // padding channels: scale = shift = 0 -> 0
          if (sbm != 1.f) v *= sbm;
          uint32_t h01, l01, h23, l23;
          split2t<F16>(v[0], v[1], h01, l01);
          split2t<F16>(v[2], v[3], h23, l23);
          // channels 16 ct + 4 g4 .. + 3: bytes 32 ct + 8 g4 of the hi half (chunk 2 ct + g4 / 2), the same of the lo half
          const int row = fl * 16 + c16;
          if constexpr (F16) {                           // rows past M hold whatever the slack behind the input held
            if (mbase + ps * ROWS + row < p.M) { ovf_bits(bad, h01); ovf_bits(bad, h23); }
          }
          char* rp = scratch + row * 128 + 8 * (g4 & 1);
          *reinterpret_cast<uint2*>(rp + (((2 * ct + (g4 >> 1)) ^ swz8(row)) << 4)) = make_uint2(h01, h23);
          *reinterpret_cast<uint2*>(rp + (((4 + 2 * ct + (g4 >> 1)) ^ swz8(row)) << 4)) = make_uint2(l01, l23);
          if constexpr (F16) __builtin_amdgcn_sched_barrier(0);   // one tile at a time: the fp16 conversions of eight
                                                                  // interleaved tiles do not fit the register budget
        }
      wave_lds_sync();
      if (ps == 0) { XV_EPI_DRAIN(); XV_EPI_STAMP(p, 6); }
#pragma unroll 4
      for (int it = 0; it < ROWS / 8; ++it) {
        const int row = it * 8 + rrow;
        f32x4 v = *reinterpret_cast<const f32x4*>(scratch + row * 128 + ((rchunk ^ swz8(row)) << 4));
        bool zero;
        const int orow = out_row_pre(ps, it, zero);
        if (zero) v = z;
        if (orow >= 0 && blk_ok)
          *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(p.Ysb) + (int64_t)orow * p.ldsb * 4 + (nbase >> 5) * 128 +
                                    rchunk * 16) = v;
      }
      if (ps == 0) XV_EPI_STAMP(p, 7);
      wave_lds_sync();
    }
  }
  if constexpr (F16) ovf_report_bits(p.ovf, bad);
  XV_EPI_STAMP(p, 5);
  if (p.Y || (ROWS == 64 && p.pool_part)) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      if (ROWS == 64 && p.pool_part && !p.Y) {
        // Fused statistics pooling, the usual pass (all 64 frames of one utterance: 3.5 passes of 4.5 at 286 frames): straight from
        // the accumulators.  A lane holds 8 channels (2 tiles x 4) of the four frames c16 + 16 fl; the 16 lanes of a DPP row hold the
        // 16 frames of a tile row, so a channel's sum over the pass is four adds in-lane and a four-step butterfly across the row
        // (quad_perm, quad_perm, row_half_mirror, row_mirror: no LDS, no wait).  Same partials as the staged form below -- (sum, M2
        // about the pass mean) per (utterance, 64-frame tile) slot -- in another summation order.
        const int mb = mbase + ps * 64;
        const int my_utt = (mb + lane < p.M) ? p.pool_row2utt[mb + lane] : -1;
        const int u_first = __builtin_amdgcn_readlane(my_utt, 0), u_last = __builtin_amdgcn_readlane(my_utt, 63);
        if (u_first >= 0 && u_first == u_last) {
          auto row_sum = [&](float x) -> float {
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));   // row_half_mirror
            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));   // row_mirror
            return x;
          };
          const int64_t slot = (int64_t)p.pool_slotbase[u_first] + (mb >> 6);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            f32x4 s1 = z;
#pragma unroll
            for (int fl = 0; fl < FPP; ++fl) s1 += value4(acc[ps * FPP + fl][ct], ct, false);
#pragma unroll
            for (int i = 0; i < 4; ++i) s1[i] = row_sum(s1[i]);
            const f32x4 mu = s1 * (1.0f / 64.0f);
            f32x4 m2 = z;
#pragma unroll
            for (int fl = 0; fl < FPP; ++fl) {
              const f32x4 d = value4(acc[ps * FPP + fl][ct], ct, false) - mu;
              m2 += d * d;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) m2[i] = row_sum(m2[i]);
            const int nn = nbase + 16 * ct + 4 * g4;
            if (c16 == 0 && nn < p.N) {
              *reinterpret_cast<f32x4*>(p.pool_part + (slot * 2) * p.N + nn) = s1;
              *reinterpret_cast<f32x4*>(p.pool_part + (slot * 2 + 1) * p.N + nn) = m2;
            }
          }
          continue;
        }
      }
      stage_f32(ps, false);
      wave_lds_sync();
      if (ps == 0) XV_EPI_STAMP(p, 6);
      if (p.Y) {
#pragma unroll 4
        for (int it = 0; it < ROWS / 8; ++it) {
          const int row = it * 8 + rrow;
          f32x4 v = *reinterpret_cast<const f32x4*>(scratch + row * 128 + ((rchunk ^ swz8(row)) << 4));
          bool zero;
          const int orow = out_row_pre(ps, it, zero);
          if (zero) v = z;
          if (orow >= 0 && n < p.N) *reinterpret_cast<f32x4*>(p.Y + (int64_t)orow * p.ldy + n) = v;
        }
      }
      if (ROWS == 64 && p.pool_part) {
        // fused statistics pooling on the staged 64 frames x 32 channels (same (sum, M2 about the segment mean)
        // partials per (utterance, 64-frame tile) slot as the 64x64 form): lane = (frame group rrow = t mod 8,
        // 4 channels rchunk); the eight frame groups are combined with three xor-shuffles.
        const int mb = mbase + ps * 64;
        const bool nok = n < p.N;
        const int my_utt = (mb + lane < p.M) ? p.pool_row2utt[mb + lane] : -1;
        const int tile64 = mb >> 6;
        auto row4 = [&](int t) -> f32x4 {
          return *reinterpret_cast<const f32x4*>(scratch + t * 128 + ((rchunk ^ swz8(t)) << 4));
        };
        auto groups_sum = [&](f32x4 v) -> f32x4 {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] += __shfl_xor(v[i], 8, 64);
            v[i] += __shfl_xor(v[i], 16, 64);
            v[i] += __shfl_xor(v[i], 32, 64);
          }
          return v;
        };
        const int u_first = __builtin_amdgcn_readlane(my_utt, 0), u_last = __builtin_amdgcn_readlane(my_utt, 63);
        if (u_first >= 0 && u_first == u_last) {
          // the usual case: all 64 frames of the pass belong to one utterance (a 286-frame utterance has a boundary
          // in one pass out of 4.5).  Fixed trip counts: the eight rows of a lane are read back to back.
          f32x4 s1 = z;
#pragma unroll 4
          for (int k = 0; k < 8; ++k) s1 += row4(k * 8 + rrow);
          s1 = groups_sum(s1);
          const f32x4 mu = s1 * (1.0f / 64.0f);
          f32x4 m2 = z;
#pragma unroll 4
          for (int k = 0; k < 8; ++k) {                 // second sweep of the LDS copy (registers are scarce here)
            const f32x4 d = row4(k * 8 + rrow) - mu;
            m2 += d * d;
          }
          m2 = groups_sum(m2);
          if (nok && rrow == 0) {
            const int64_t slot = (int64_t)p.pool_slotbase[u_first] + tile64;
            *reinterpret_cast<f32x4*>(p.pool_part + (slot * 2) * p.N + n) = s1;
            *reinterpret_cast<f32x4*>(p.pool_part + (slot * 2 + 1) * p.N + n) = m2;
          }
        } else {
          // utterance boundaries (or the end of the matrix) inside the pass: one (sum, M2) pair per segment; the segment
          // ends come from one ballot per segment instead of a lane-by-lane scan
          int r = 0;
          while (r < 64) {
            const int b = __builtin_amdgcn_readlane(my_utt, r);
            const unsigned long long same = __ballot(my_utt == b) >> r;          // bit 0 = lane r (set)
            const int re = r + (~same ? __builtin_ctzll(~same) : 64 - r);
            if (b >= 0) {
              f32x4 s1 = z;
              for (int t0 = r & ~7; t0 < re; t0 += 8) {
                const int t = t0 + rrow;
                if (t >= r && t < re) s1 += row4(t);
              }
              s1 = groups_sum(s1);
              const f32x4 mu = s1 / (float)(re - r);
              f32x4 m2 = z;
              for (int t0 = r & ~7; t0 < re; t0 += 8) {
                const int t = t0 + rrow;
                if (t >= r && t < re) {
                  const f32x4 d = row4(t) - mu;
                  m2 += d * d;
                }
              }
              m2 = groups_sum(m2);
              if (nok && rrow == 0) {
                const int64_t slot = (int64_t)p.pool_slotbase[b] + tile64;
                *reinterpret_cast<f32x4*>(p.pool_part + (slot * 2) * p.N + n) = s1;
                *reinterpret_cast<f32x4*>(p.pool_part + (slot * 2 + 1) * p.N + n) = m2;
              }
            }
            r = re;
          }
        }
      }
      if (ps == 0) XV_EPI_STAMP(p, 7);
      wave_lds_sync();
    }
  }
}

// Attention epilogue of the 128-frame x 32-channel wave tile (kernel instantiations with EPI = 1; dense layers only:
// no rowmap).  Two forms, see GemmArgs:
//  * att_part: the tile is the last key layer's output and only its dot products with the query are wanted
//    (model/pooling.py:189-194).  A lane holds 8 channels of one frame per 16-frame tile row, so the partial score of
//    (frame, head) is 8 FMAs in-lane plus two exchanges across the four lane groups; the 16 frames of a tile
//    are stored as one 64-byte run.  The 1500-wide key never leaves the registers.
//  * pool_w: the tile is the value and only its weighted moments are wanted (:201-217): 64 frames are staged in the
//    wave-private LDS scratch as in the statistics form, then per utterance segment and head s1 = sum w x and
//    m2 = sum w (x - s1 / s0)^2 go to the segment's slot.
template <int ACT, int FORM>
__device__ __forceinline__ void store_wave_tile_n32_att_impl(const GemmArgs& p, const f32x4 (&acc)[8][2], int mbase, int nbase,
                                                             int lane, int wave, char* lds) {
  const int c16 = lane & 15, g4 = lane >> 4;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 sc[2], sh[2];                                   // this lane's channels 16 ct + 4 g4 .. +3, ct = 0, 1
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int n4 = nbase + 16 * ct + 4 * g4;
    const bool ok = n4 < p.N;
    sc[ct] = ok ? *reinterpret_cast<const f32x4*>(p.scale + n4) : z;
    sh[ct] = ok ? *reinterpret_cast<const f32x4*>(p.shift + n4) : z;
  }
  auto value4 = [&](const f32x4& t, int ct) -> f32x4 {
    f32x4 al = z;                                        // PReLU slopes: loaded at the point of use (registers are scarce)
    if (ACT < 0 && p.alpha && nbase + 16 * ct + 4 * g4 < p.N) al = *reinterpret_cast<const f32x4*>(p.alpha + nbase + 16 * ct + 4 * g4);
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = apply_act(fmaf(t[i], sc[ct][i], sh[ct][i]), ACT < 0 ? p.act : ACT, al[i]);
    return v;
  };
  if constexpr (FORM == 1) {
    const int blk = nbase >> 5;
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) {
      f32x4 v[2];                                        // this lane's 8 channels of frame ft * 16 + c16, activated
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) v[ct] = value4(acc[ft][ct], ct);   // padding channels: scale = shift = 0 and query 0
      const int m = mbase + ft * 16 + c16;
      for (int hd = 0; hd < p.att_heads; ++hd) {
        const float* qp = p.att_q + (int64_t)hd * p.Npad + nbase + 4 * g4;
        float s = 0.f;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const f32x4 qv = *reinterpret_cast<const f32x4*>(qp + 16 * ct);
#pragma unroll
          for (int i = 0; i < 4; ++i) s = fmaf(v[ct][i], qv[i], s);
        }
        s += __shfl_xor(s, 16, 64);                      // the other 24 channels of this frame (lane groups g4)
        s += __shfl_xor(s, 32, 64);
        if (g4 == 0 && m < p.M) p.att_part[((int64_t)blk * p.att_heads + hd) * p.att_ld + m] = s;
      }
    }
    return;
  } else {
  // ---- weighted moments of the value
  char* scratch = lds + wave * (64 * 128);
  const int rrow = lane >> 3, rchunk = lane & 7;        // read-back map: 8 frames x 8 chunks per pass
  const int n = nbase + rchunk * 4;
  const bool nok = n < p.N;
  const int H = p.pool_heads;
  const int hd0 = p.pool_split ? min(nbase / p.pool_dvh, H - 1) : 0;     // split value: one head per 32-channel block
  const int hd1 = p.pool_split ? hd0 + 1 : H;
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
    for (int fl = 0; fl < 4; ++fl)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int row = fl * 16 + c16;
        *reinterpret_cast<f32x4*>(scratch + row * 128 + (((4 * ct + g4) ^ swz8(row)) << 4)) = value4(acc[ps * 4 + fl][ct], ct);
      }
    wave_lds_sync();
    const int mb = mbase + ps * 64;
    const int my_utt = (mb + lane < p.M) ? p.pool_row2utt[mb + lane] : -1;
    const int tile64 = mb >> 6;
    auto row4 = [&](int t) -> f32x4 {
      return *reinterpret_cast<const f32x4*>(scratch + t * 128 + ((rchunk ^ swz8(t)) << 4));
    };
    auto groups_sum = [&](f32x4 v) -> f32x4 {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] += __shfl_xor(v[i], 8, 64);
        v[i] += __shfl_xor(v[i], 16, 64);
        v[i] += __shfl_xor(v[i], 32, 64);
      }
      return v;
    };
    for (int hd = hd0; hd < hd1; ++hd) {
      const float wl = (mb + lane < p.M) ? p.pool_w[(int64_t)(mb + lane) * H + hd] : 0.f;   // lane = frame of this pass
      const int oc = p.pool_split ? n : hd * p.N + n;
      const int u_first = __builtin_amdgcn_readlane(my_utt, 0), u_last = __builtin_amdgcn_readlane(my_utt, 63);
      if (u_first >= 0 && u_first == u_last) {
        // all 64 frames of the pass belong to one utterance: fixed trip counts
        float s0 = wl;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s0 += __shfl_xor(s0, o, 64);
        float wt[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) wt[k] = __shfl(wl, k * 8 + rrow, 64);
        f32x4 s1 = z;
#pragma unroll 4
        for (int k = 0; k < 8; ++k) s1 += row4(k * 8 + rrow) * wt[k];
        s1 = groups_sum(s1);
        const f32x4 mu = s1 * (s0 > 0.f ? 1.0f / s0 : 0.f);
        f32x4 m2 = z;
#pragma unroll 4
        for (int k = 0; k < 8; ++k) {                   // second sweep of the LDS copy (registers are scarce here)
          const f32x4 d = row4(k * 8 + rrow) - mu;
          m2 += d * d * wt[k];
        }
        m2 = groups_sum(m2);
        if (nok && rrow == 0) {
          const int64_t slot = (int64_t)p.pool_slotbase[u_first] + tile64;
          *reinterpret_cast<f32x4*>(p.pool_part + (slot * 2) * p.pool_odim + oc) = s1;
          *reinterpret_cast<f32x4*>(p.pool_part + (slot * 2 + 1) * p.pool_odim + oc) = m2;
        }
      } else {
        int r = 0;
        while (r < 64) {
          const int b = __builtin_amdgcn_readlane(my_utt, r);
          const unsigned long long same = __ballot(my_utt == b) >> r;          // bit 0 = lane r (set)
          const int re = r + (~same ? __builtin_ctzll(~same) : 64 - r);
          if (b >= 0) {
            float s0 = (lane >= r && lane < re) ? wl : 0.f;           // sum of the segment's weights
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s0 += __shfl_xor(s0, o, 64);
            f32x4 s1 = z;
            for (int t0 = r & ~7; t0 < re; t0 += 8) {
              const int t = t0 + rrow;
              const float wt = __shfl(wl, t & 63, 64);
              if (t >= r && t < re) s1 += row4(t) * wt;
            }
            s1 = groups_sum(s1);
            const f32x4 mu = s1 * (s0 > 0.f ? 1.0f / s0 : 0.f);
            f32x4 m2 = z;
            for (int t0 = r & ~7; t0 < re; t0 += 8) {
              const int t = t0 + rrow;
              const float wt = __shfl(wl, t & 63, 64);
              if (t >= r && t < re) {
                const f32x4 d = row4(t) - mu;
                m2 += d * d * wt;
              }
            }
            m2 = groups_sum(m2);
            if (nok && rrow == 0) {
              const int64_t slot = (int64_t)p.pool_slotbase[b] + tile64;
              *reinterpret_cast<f32x4*>(p.pool_part + (slot * 2) * p.pool_odim + oc) = s1;
              *reinterpret_cast<f32x4*>(p.pool_part + (slot * 2 + 1) * p.pool_odim + oc) = m2;
            }
          }
          r = re;
        }
      }
    }
    wave_lds_sync();
  }
  }
}

// FORM 1 = score partials (GemmArgs::att_part), 2 = weighted moments (GemmArgs::pool_w)
template <int FORM>
__device__ __forceinline__ void store_wave_tile_n32_att(const GemmArgs& p, const f32x4 (&acc)[8][2], int mbase, int nbase,
                                                        int lane, int wave, char* lds) {
  if (p.act == ACT_RELU)
    store_wave_tile_n32_att_impl<ACT_RELU, FORM>(p, acc, mbase, nbase, lane, wave, lds);
  else
    store_wave_tile_n32_att_impl<-1, FORM>(p, acc, mbase, nbase, lane, wave, lds);
}

// Epilogue entry for the 128x32 wave tile.
template <int ROWS, bool F16>
__device__ __forceinline__ void store_wave_tile_n32(const GemmArgs& p, const f32x4 (&acc)[8][2], int mbase, int nbase,
                                                    int lane, int wave, char* lds);

// true when the vectorised LDS-staged epilogue applies to this launch
__device__ __forceinline__ bool wide_epilogue_ok(const GemmArgs& p) {
  return (p.N & 3) == 0 && (!p.Y || ((p.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(p.Y) & 15) == 0)) &&
         (!p.pool_part || p.rowmap == nullptr) &&
         (!p.R || ((p.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(p.R) & 15) == 0));
}

// Epilogue entry for a 64x64 wave tile: LDS-staged wide stores when the shape allows, scalar otherwise.
// ROWS = 32 must not be used for a launch with pool_part (the fused pooling needs the whole tile staged).
template <int ROWS = 64, int MI = 2>
__device__ __forceinline__ void store_wave_tile(const GemmArgs& p, const f32x16 (&acc)[2][MI], int mbase, int nbase,
                                                int lane, int wave, char* lds) {
  if (wide_epilogue_ok(p)) {
    if (p.act == ACT_RELU)
      store_wave_tile_impl<ACT_RELU, ROWS, MI>(p, acc, mbase, nbase, lane, wave, lds);
    else if (p.act == ACT_NONE)
      store_wave_tile_impl<ACT_NONE, ROWS, MI>(p, acc, mbase, nbase, lane, wave, lds);
    else
      store_wave_tile_impl<-1, ROWS, MI>(p, acc, mbase, nbase, lane, wave, lds);
    return;
  }
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) store_tile_scalar(p, acc[ni][mi], mbase + mi * 32, nbase + ni * 32, lane);
}

template <int ROWS, bool F16>
__device__ __forceinline__ void store_wave_tile_n32(const GemmArgs& p, const f32x4 (&acc)[8][2], int mbase, int nbase,
                                                    int lane, int wave, char* lds) {
  if (wide_epilogue_ok(p)) {
    if (p.act == ACT_RELU)
      store_wave_tile_n32_impl<ACT_RELU, ROWS, F16>(p, acc, mbase, nbase, lane, wave, lds);
    else if (p.act == ACT_NONE)
      store_wave_tile_n32_impl<ACT_NONE, ROWS, F16>(p, acc, mbase, nbase, lane, wave, lds);
    else
      store_wave_tile_n32_impl<-1, ROWS, F16>(p, acc, mbase, nbase, lane, wave, lds);
    return;
  }
#pragma unroll
  for (int ft = 0; ft < 8; ++ft)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) store_tile16_scalar(p, acc[ft][ct], mbase + ft * 16, nbase + ct * 16, lane);
}

}  // namespace xv
